import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import scenarios as S
from tests import gpu_util as U, oracle_lib as O
from pipeinfer_amd import llama as LL
U.init()
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
stem, seed = "tiny_llama_b_q4_k_m", 13
path = os.path.join(GOLD, stem + ".gguf")
m = LL.Model(path); c = LL.Context(m, n_ctx=128, n_batch=64)
om = O.OracleModel(path, n_ctx=128, max_seq=8)
prompt, n_gen = S.scenario_greedy(m.n_vocab, seed=seed)
lg = c.decode(prompt, list(range(len(prompt))), [[0]] * len(prompt))
lo = om.decode(prompt, list(range(len(prompt))), [[0]] * len(prompt))
cur = int(np.argmax(lo[-1]))
for i in range(13):
    if i == 12:
        os.environ["PI_TRACE"] = "1"
        sys.stderr.write("ENGINE\n"); sys.stderr.flush()
    lg = c.decode([cur], [len(prompt) + i], [[0]])
    if i == 12:
        sys.stderr.write("ORACLE\n"); sys.stderr.flush()
    lo = om.decode([cur], [len(prompt) + i], [[0]])
    cur = int(np.argmax(lo[0]))
