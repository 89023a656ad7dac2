"""Brute-force the XOR swizzles of the chained 1x1 kernel's LDS images (csrc/chain.hip): every ds_read_b128 lane group must hit 16 distinct 16-byte slots
of the 256-byte bank row (MI355X_MICROARCH.md, LDS table)."""
import itertools

GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
          list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
          list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
          list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]


def ok(addr_of_lane):
    for g in GROUPS:
        slots = {}
        for l in g:
            a = addr_of_lane(l)
            s = (a // 16) % 16
            if s in slots and slots[s] != a:
                return False
            slots[s] = a
    return True


def wrow(frow, base):
    return base + 8 * (frow >> 2) + (frow & 3)


def check_w(rowbytes, f, bases, ksteps):
    for base in bases:
        for ks in range(ksteps):
            def addr(l, base=base, ks=ks):
                frow, q = l & 15, l >> 4
                r = wrow(frow, base)
                u = 4 * ks + q
                return r * rowbytes + 16 * (u ^ f(r))
            if not ok(addr):
                return False
    return True


cands = {}
for a in range(0, 6):
    for m in (1, 3, 7, 15):
        cands["(r>>%d)&%d" % (a, m)] = (lambda r, a=a, m=m: (r >> a) & m)
        for b in range(0, 6):
            for sh in (1, 2, 3):
                cands["((r>>%d)&%d) ^ (((r>>%d)&1)<<%d)" % (a, m, b, sh)] = (lambda r, a=a, m=m, b=b, sh=sh: ((r >> a) & m) ^ (((r >> b) & 1) << sh))

# W3 half block: 64 rows x 256 B, tiles t=0..3 -> base 32*(t>>1) + 4*(t&1); 4 k-steps
print("W3 (256-B rows):")
for name, f in cands.items():
    if all(f(r) < 16 for r in range(64)) and check_w(256, f, [32 * (t >> 1) + 4 * (t & 1) for t in range(4)], 4):
        print("  ", name)
# W1 half block: 128 rows x 128 B, tiles i=0..7 -> base 64*(i>>2) + 32*((i>>1)&1) + 4*(i&1); 2 k-steps
print("W1 (128-B rows):")
for name, f in cands.items():
    if all(f(r) < 8 for r in range(128)) and check_w(128, f, [64 * (i >> 2) + 32 * ((i >> 1) & 1) + 4 * (i & 1) for i in range(8)], 2):
        print("  ", name)
# residual / X1 image: rows m (16 per m-tile) x 128 B, lane (m = l & 15, q): slot u = 4 s + q
print("RES (128-B rows, row = lane & 15):")
for name, f in cands.items():
    if not all(f(r) < 8 for r in range(64)):
        continue
    good = True
    for s in range(2):
        def addr(l, s=s):
            m, q = l & 15, l >> 4
            return m * 128 + 16 * ((4 * s + q) ^ f(m))
        if not ok(addr):
            good = False
    if good:
        print("  ", name)

# ---- the 8-wave version's blocks -------------------------------------------------------------------------------------------
# first product: 32 rows (two MFMA tiles t = 0, 1 -> base 4 t) x 512 B (all of K1 = 256), 8 k-steps
print("A block (512-B rows, 32 rows):")
for name, f in cands.items():
    if all(f(r) < 16 for r in range(32)) and check_w(512, f, [0, 4], 8):
        print("  ", name)
# second product: 256 rows x 64 B (one 32-wide k-step), 16 tiles, slot u = q
print("B block (64-B rows, 256 rows):")
for name, f in cands.items():
    if all(f(r) < 4 for r in range(256)) and check_w(64, f, [64 * (i >> 2) + 32 * ((i >> 1) & 1) + 4 * (i & 1) for i in range(16)], 1):
        print("  ", name)
