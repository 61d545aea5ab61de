/* Exhaustive check: lz_expf / lz_logf (lightzero_amd/csrc/lz_math.h) == host libm expf / logf for
 * every binary32 input (or a strided subset: argv[1] = stride).  Prints mismatch counts. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "../lightzero_amd/csrc/lz_math.h"

int main(int argc, char **argv)
{
    unsigned long stride = argc > 1 ? strtoul(argv[1], 0, 10) : 1;
    unsigned long bad_exp = 0, bad_log = 0, n = 0;
    unsigned first_exp = 0, first_log = 0;
#pragma omp parallel for reduction(+ : bad_exp, bad_log, n) schedule(static)
    for (unsigned long u = 0; u < (1ul << 32); u += stride) {
        float x = lz_asfloat((uint32_t)u);
        float a = lz_expf(x), b = expf(x);
        if (lz_asuint(a) != lz_asuint(b) && !(a != a && b != b)) { if (!bad_exp) first_exp = (unsigned)u; bad_exp++; }
        a = lz_logf(x); b = logf(x);
        if (lz_asuint(a) != lz_asuint(b) && !(a != a && b != b)) { if (!bad_log) first_log = (unsigned)u; bad_log++; }
        n++;
    }
    printf("checked %lu inputs: expf mismatches %lu (first 0x%08x), logf mismatches %lu (first 0x%08x)\n", n, bad_exp,
           first_exp, bad_log, first_log);
    return (bad_exp || bad_log) ? 1 : 0;
}
