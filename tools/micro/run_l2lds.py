"""Per-CU delivery rate of an operand tile into LDS (tools/micro/l2lds.hip): segment width x path x cache level."""
import ctypes, os, torch
here = os.path.dirname(os.path.abspath(__file__))
L = ctypes.CDLL(os.path.join(here, "l2lds.so"))
L.run_l2lds.argtypes = [ctypes.c_void_p, ctypes.c_size_t] + [ctypes.c_int] * 7 + [ctypes.c_void_p, ctypes.c_void_p]
src = torch.randint(0, 255, (3 << 30,), dtype=torch.uint8, device="cuda")
sink = torch.zeros(4096, dtype=torch.int32, device="cuda")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
ITERS = 3072            # stages of 16 KB per workgroup

# shader clock under load: s_memtime ticks per microsecond of s_memrealtime (100 MHz)
L.run_clock_probe.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
probe = torch.zeros(1024, dtype=torch.int32, device="cuda")
for name, mfma, iters, grid, threads in (("MFMA 16x16x32 bf16 back to back, 1 wave / SIMD, every CU", 1, 40000, 256, 256),
                                         ("MFMA 16x16x32 bf16 back to back, 2 waves / SIMD, every CU", 1, 40000, 256, 512),
                                         ("MFMA 16x16x32 bf16 back to back, 4 waves / SIMD, every CU", 1, 20000, 256, 1024),
                                         ("MFMA 32x32x16 bf16 back to back, 1 wave / SIMD, every CU", 2, 40000, 256, 256),
                                         ("MFMA 32x32x16 bf16 back to back, 2 waves / SIMD, every CU", 2, 40000, 256, 512),
                                         ("MFMA 16x16x32, 2 waves / SIMD, one CU only", 1, 40000, 1, 512),
                                         ("LDS-DMA stream from L2, 2 waves / SIMD, every CU", 0, 20000, 256, 512)):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for rep in range(2):
        e0.record()
        assert L.run_clock_probe(src.data_ptr(), mfma, iters, grid, threads, probe.data_ptr(), st) == 0
        e1.record()
        torch.cuda.synchronize()
    t = probe[:2].tolist()
    waves = threads // 64
    per_iter = 8 if mfma == 1 else 4
    extra = "  %5.1f ticks per MFMA per SIMD, %.0f TFLOP/s chip by the event time" % (t[0] / (iters * per_iter * waves / 4), 2 * 16 * 16 * 32 * per_iter * (2 if mfma == 2 else 1) * iters * waves * grid / (e0.elapsed_time(e1) * 1e-3) / 1e12) if mfma else ""
    print("clock probe, %-60s: %9d ticks in %8.2f us (events: %8.2f us) = %.3f GHz%s" % (name, t[0], t[1] / 100.0, e0.elapsed_time(e1) * 1e3, t[0] / (t[1] * 10.0), extra))
if os.environ.get("CLOCK_ONLY"):
    raise SystemExit(0)


def run(mode, barrier, stride, seg, blocks, grid):
    def go():
        rc = L.run_l2lds(src.data_ptr(), src.numel(), mode, barrier, stride, seg, blocks, ITERS, grid, sink.data_ptr(), st)
        assert rc == 0, rc
    go(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        go()
    e1.record(); torch.cuda.synchronize()
    sec = e0.elapsed_time(e1) / 3 * 1e-3
    return ITERS * 16384 * grid / sec           # bytes/s chip


for grid in (256, 512):
    print("grid %d (%d workgroup(s) of 512 threads per CU), 16 KB stages, 3 in flight; GB/s per CU | TB/s chip" % (grid, grid // 256))
    for level, blocks_of in (("L2 (64 KB/WG)", lambda bb: max(1, 65536 // bb)), ("MALL (512 KB/WG)", lambda bb: max(1, (512 << 10) // bb)),
                             ("HBM (8 MB/WG)" if grid == 256 else "HBM (4 MB/WG)", lambda bb: max(1, ((8 << 20) if grid == 256 else (4 << 20)) // bb))):
        for stride, seg in ((128, 64), (512, 64), (2048, 64), (256, 128), (512, 128), (2048, 128), (512, 256), (2048, 256), (512, 512), (2048, 1024)):
            bb = (512 // (seg // 16)) * 2 * stride          # block bytes
            blocks = blocks_of(bb)
            row = []
            for mode, barrier in ((0, 0), (0, 1), (1, 0), (2, 0)):
                bw = run(mode, barrier, stride, seg, blocks, grid)
                row.append("%6.1f|%5.2f" % (bw / 256 / 1e9, bw / 1e12))
            print("%-18s stride %4d seg %4d region %6d KB | dma %s | dma+barrier %s | vgpr+ds_write %s | vgpr only %s" % (
                level, stride, seg, bb * blocks >> 10, *row))
