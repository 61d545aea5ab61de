#include "lz_internal.h"
struct lz_model { int dummy; };
void lz_model_destroy(lz_model *m) { delete m; }
