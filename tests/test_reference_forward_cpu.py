"""CPU: tests/reference_forward_collect.py claims to BE the reference's EfficientZeroPolicy._forward_collect / _forward_eval with only
the import lines changed.  Where /root/reference exists this is checked statement by statement: the AST of each restated method
(docstring dropped) must equal the AST of the reference's method."""
import ast
import os

import pytest

REF = "/root/reference/lzero/policy/efficientzero.py"
HERE = os.path.dirname(os.path.abspath(__file__))


def _methods(path, cls):
    tree = ast.parse(open(path).read())
    out = {}
    for c in tree.body:
        if isinstance(c, ast.ClassDef) and c.name == cls:
            for f in c.body:
                if isinstance(f, ast.FunctionDef):
                    body = f.body
                    if body and isinstance(body[0], ast.Expr) and isinstance(getattr(body[0], "value", None), ast.Constant):
                        body = body[1:]
                    out[f.name] = (ast.dump(f.args), [ast.dump(b) for b in body])
    return out


@pytest.mark.skipif(not os.path.isfile(REF), reason="/root/reference is not on this machine")
@pytest.mark.parametrize("name", ["_forward_collect", "_forward_eval"])
def test_restated_forward_is_the_reference_statement_by_statement(name):
    ref = _methods(REF, "EfficientZeroPolicy")[name]
    mine = _methods(os.path.join(HERE, "reference_forward_collect.py"), "ReferenceForwardBodies")[name]
    assert mine[0] == ref[0], "signature differs"
    assert len(mine[1]) == len(ref[1])
    for k, (a, b) in enumerate(zip(mine[1], ref[1])):
        assert a == b, "statement %d of %s differs from the reference" % (k, name)


def test_only_the_import_lines_point_at_the_drop_in():
    src = open(os.path.join(HERE, "reference_forward_collect.py")).read()
    imports = [l for l in src.splitlines() if l.startswith(("from ", "import "))]
    assert [l.split("#")[0].strip() for l in imports if "lightzero_amd" in l] == [
        "from lightzero_amd.mcts.tree_search.mcts_ctree import EfficientZeroMCTSCtree as MCTSCtree",
        "from lightzero_amd.policy.utils import select_action, ez_network_output_unpack"]
    body = src.split('way _init_collect / _init_eval do"""', 1)[1]
    assert "oracle" not in body and "lightzero_amd" not in body and "lzero" not in body
