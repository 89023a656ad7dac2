"""The s_waitcnt vmcnt(N) constants of csrc/chain.hip, from a replay of one wave's vector-memory operations in program order.
A wait "for X" may leave outstanding exactly the operations issued AFTER the last operation of X (vmcnt retires in order).
Program of a wave per chunk (PW weight pieces per block and wave, PR residual pieces, PS store instructions per chunk):
  READ(A0): W(+4) R(c+2) | g1: wait(A1)    MFMA(A0): waitR(c) epi0 | g0: wait(A1)
  READ(A1): W(+4)        | g1: wait(B0)    MFMA(A1): epi1 S(c)     | g0: wait(B0)
  READ(B0): W(+4)        | g1: wait(B1)    MFMA(B0):               | g0: wait(B1)
  READ(B1): W(+4)        | g1: wait(A0')   MFMA(B1):               | g0: wait(A0')
prologue: W(0) W(1) W(2) W(3) R(0) R(1), wait(block 0)."""
import sys

PW, PR, PS = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (2, 2, 2)
NCHUNK = 8
ops = []          # (kind, id) per issued operation


def issue(kind, ident, n):
    for _ in range(n):
        ops.append((kind, ident))


def younger(kind, ident):
    last = max(i for i, o in enumerate(ops) if o == (kind, ident))
    return len(ops) - 1 - last


res = {}


def note(key, val):
    res.setdefault(key, []).append(val)


for b in range(4):
    issue("W", b, PW)
issue("R", 0, PR)
issue("R", 1, PR)
note("prologue wait(block 0)", younger("W", 0))
for c in range(NCHUNK):
    for pos, name in enumerate(("A0", "A1", "B0", "B1")):
        x = 4 * c + pos
        issue("W", x + 4, PW)
        if pos == 0:
            issue("R", c + 2, PR)
        note("g1 end of READ(%s): wait(next block)" % name, younger("W", x + 1))
        if pos == 0:
            note("MFMA(A0): waitR(c), chunk %s" % (c if c < 2 else "steady"), younger("R", c))
        if pos == 1:
            issue("S", c, PS)
        note("g0 end of MFMA(%s): wait(next block)" % name, younger("W", x + 1))
for k, v in res.items():
    print("%-48s %s" % (k, v[:3] + ["..."] + v[-2:] if len(v) > 5 else v))
