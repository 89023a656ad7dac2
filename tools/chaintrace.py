"""s_memtime timeline of one workgroup of the chained 1x1 kernel (csrc/chain.hip): `python tools/chaintrace.py build [pass] [-DCHAIN_DBG=..]` links a second
library with -DCHAIN_TRACE=<pass>, `python tools/chaintrace.py [bwd]` runs one launch at the step's size and prints the cycles between the sixteen stamps of
a chunk (median over chunks 2 .. 15) and the shader clock the pass ran at."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.environ.get("CHAIN_TRACE_LIB") or os.path.join(ROOT, "tools", "experiments", "libchain_trace.so")
NAMES = ["A0: read 2nd half + MFMA 1st half", "A0: wait + barrier", "A0: issue W,R + read next 1st half", "A0: MFMA 2nd half + residual wait + epilogue half 0",
         "A1: read + MFMA 1st half", "A1: wait + barrier", "A1: issue W + read", "A1: MFMA 2nd half + epilogue half 1 + stores",
         "B0: read + MFMA 1st half", "B0: wait + barrier", "B0: issue W + read", "B0: MFMA 2nd half",
         "B1: read + MFMA 1st half", "B1: wait + barrier", "B1: issue W + read", "B1: MFMA 2nd half"]

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
buf = (ctypes.c_uint * 516)()
assert h.mi_chain_trace_read(buf, 516) == 0
for grp in range(2):
    t = np.array(buf[grp * 256:grp * 256 + 256], dtype=np.int64).reshape(16, 16)
    d = np.zeros((16, 16), dtype=np.int64)
    d[:, :15] = t[:, 1:] - t[:, :15]
    d[:15, 15] = t[1:, 0] - t[:15, 15]
    d &= 0xFFFFFFFF
    print("group %d: cycles between stamps, median / min / max over chunks 2..14 (one wave; %s)" % (grp, "backward" if bwd else "forward"))
    for k in range(16):
        col = d[2:15, k]
        print("  %2d %-56s %6d %6d %6d" % (k, NAMES[k], np.median(col), col.min(), col.max()))
    print("  chunk total (median): %d cycles" % np.median(d[2:15].sum(1)))
    cyc, rt = buf[512 + 2 * grp], buf[513 + 2 * grp]
    print("  pass: %d cycles, %.1f us -> clock %.2f GHz" % (cyc, rt / 100.0, cyc / (rt * 10.0)))
