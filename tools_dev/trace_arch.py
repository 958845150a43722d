"""first divergence between the engine and the oracle on one fixture: both print per-operator sums with PI_TRACE=1 (not a test)
usage: PI_TRACE=1 python tools_dev/trace_arch.py <stem> 2> trace.txt"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pipeinfer_amd import llama as LL
from tests import gpu_util as U, oracle_lib as O
U.init()
stem = sys.argv[1]
path = os.path.join(ROOT, "tests", "golden", stem + ".gguf")
toks, pos, seqs = [5, 9, 100, 3], [0, 1, 2, 3], [[0]] * 4
sys.stderr.write("== oracle\n"); sys.stderr.flush()
m = O.OracleModel(path, n_ctx=64, max_seq=4)
a = m.decode(toks, pos, seqs); m.close()
sys.stderr.write("== engine\n"); sys.stderr.flush()
mm = LL.Model(path); c = LL.Context(mm, n_ctx=64, n_batch=16)
b = c.decode(toks, pos, seqs)
print(stem, "max |engine - oracle| =", float(np.abs(a - b).max()))
