"""s_memtime timeline of one workgroup of the chained 1x1 kernel (csrc/chain.hip): `python tools/chaintrace.py build [pass] [-DCHAIN_DBG=..]` links a second
library with -DCHAIN_TRACE=<pass>, `python tools/chaintrace.py [bwd]` runs one launch at the step's size and prints the cycles between the sixteen stamps of
a chunk (median over chunks 2 .. 15) and the shader clock the pass ran at."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.environ.get("CHAIN_TRACE_LIB") or os.path.join(ROOT, "tools", "experiments", "libchain_trace.so")
NAMES = ["top -> A0 first half", "wait + barrier", "issue W,R + read A1", "A0 second + A1 first half", "wait + barrier", "issue W + read B0", "A1 second half",
         "residual wait", "epilogue + stores", "B0 first half", "wait + barrier", "issue W + read B1", "B0 second + B1 first half", "wait + barrier",
         "issue W + read A0'", "B1 second half -> next top"]

if len(sys.argv) > 1 and sys.argv[1] == "build":
    ps = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else "0"
    extra = " ".join(a for a in sys.argv[2:] if a.startswith("-"))
    subprocess.run([os.path.join(ROOT, "tools", "dbg", "chain_variants.sh"), "trace:-DCHAIN_TRACE=%s %s" % (ps, extra)], check=True)
    sys.exit(0)

os.environ["MI355SEG_LIB"] = LIB
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from rnd_semantic_segmentation_amd import _lib, kernels as K  # noqa: E402
import test_gpu_chain as T  # noqa: E402

T.K = K
bwd = len(sys.argv) > 1 and sys.argv[1] == "bwd"
ops = T._operands(8, 97, 97, 3, bwd)
for _ in range(3):
    T._chain(ops, bwd)
torch.cuda.synchronize()
h = ctypes.CDLL(LIB)
buf = (ctypes.c_uint * 260)()
assert h.mi_chain_trace_read(buf, 260) == 0
t = np.array(buf[:256], dtype=np.int64).reshape(16, 16)
d = np.zeros((16, 16), dtype=np.int64)
d[:, :15] = t[:, 1:] - t[:, :15]
d[:15, 15] = t[1:, 0] - t[:15, 15]
d &= 0xFFFFFFFF
print("cycles between stamps, median / min / max over chunks 2..14 (one wave; %s)" % ("backward" if bwd else "forward"))
for k in range(16):
    col = d[2:15, k]
    print("  %2d %-32s %6d %6d %6d" % (k, NAMES[k], np.median(col), col.min(), col.max()))
print("  chunk total (median): %d cycles" % np.median(d[2:15].sum(1)))
print("  pass: %d cycles, %.1f us -> clock %.2f GHz" % (buf[256], buf[257] / 100.0, buf[256] / (buf[257] * 10.0)))
