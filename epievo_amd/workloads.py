"""The synthetic workloads of BASELINE.json (inputs of bench.py, the tools and the tests).

The reference ships two tiny input files, test/test.param and test/tree.nwk; their 2 + 1 lines
are restated here as data so that the benchmark needs nothing outside the package.  The
16-leaf balanced tree of configuration 5 is not in the reference: all branch lengths 0.05
(SURVEY.md section 8d)."""
import os
import tempfile

from . import host

TEST_PARAM_TEXT = "stationary\t0.844912\t0.893359\nbaseline\t-0.8\t-1.8\n"
TREE_NWK_TEXT = "((C:0.03,D:0.06)E:0.02,F:0.1)G:0.0;\n"
# further topologies for the parity tests: a star (root with four children), a tree with an
# internal trifurcation and one long branch, and a 6-leaf caterpillar
EXTRA_TREES = {
    "star4": "(A:0.1,B:0.2,C:0.05,D:0.3)R:0.0;\n",
    "multi": "((A:0.1,B:0.1,C:0.2)X:0.1,(D:0.3,E:0.05)Y:0.2,F:0.8)R:0.0;\n",
    "cat6": "(((((A:0.05,B:0.07)U:0.04,C:0.1)V:0.03,D:0.15)W:0.06,E:0.2)Z:0.02,F:0.25)R:0.0;\n",
}


def _tmp(name, text):
    d = os.path.join(tempfile.gettempdir(), "epv_inputs_%d" % os.getuid())
    os.makedirs(d, exist_ok=True)
    p = os.path.join(d, name)
    # several rank processes write the same file: never expose a truncated one
    try:
        if open(p).read() == text:
            return p
    except OSError:
        pass
    tmp = "%s.%d.tmp" % (p, os.getpid())
    with open(tmp, "w") as f:
        f.write(text)
    os.replace(tmp, p)
    return p


def ref_test_model():
    return host.Model.read(_tmp("test.param", TEST_PARAM_TEXT), scale=True)


def tree_nwk():
    return host.Tree.read(_tmp("tree.nwk", TREE_NWK_TEXT))


def config(name):
    """the tree of the named configuration: tree (4-leaf test/tree.nwk), pair (one branch,
    T = 1), bal16 / bal32 (balanced), cat20 (caterpillar), or one of EXTRA_TREES"""
    if name == "tree":
        return tree_nwk()
    if name == "pair":
        return host.Tree.single_branch(1.0)
    if name == "bal16":
        return host.Tree.balanced(16, 0.05)
    if name == "bal8":           # 15 nodes: between the trees the LDS record pool was made for and the large-tree kernels
        return host.Tree.balanced(8, 0.05)
    if name == "bal32":          # 63 nodes: the widest tree the large-tree kernels take (node masks are one word)
        return host.Tree.balanced(32, 0.03)
    if name == "bal64":          # 127 nodes: two words per node mask in the large-tree kernels
        return host.Tree.balanced(64, 0.02)
    if name == "cat20":          # a caterpillar of 20 leaves: 39 nodes, 19 levels
        text = "(L0:0.05,L1:0.07)I0:0.03"
        for i in range(2, 20):
            text = "(%s,L%d:%.3f)I%d:%.3f" % (text, i, 0.04 + 0.01 * (i % 5), i - 1, 0.02 + 0.005 * (i % 3))
        return host.Tree.read(_tmp("cat20.nwk", text.rsplit(":", 1)[0] + ":0.0;\n"))
    if name in EXTRA_TREES:
        return host.Tree.read(_tmp(name + ".nwk", EXTRA_TREES[name]))
    raise KeyError(name)


def simulate(name, n, seed=42):
    m = ref_test_model()
    t = config(name)
    return m, t, host.simulate(m, t, n, seed)
