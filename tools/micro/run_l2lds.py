"""Per-CU delivery rate of an operand tile into LDS (tools/micro/l2lds.hip): segment width x path x cache level."""
import ctypes, os, torch
here = os.path.dirname(os.path.abspath(__file__))
L = ctypes.CDLL(os.path.join(here, "l2lds.so"))
L.run_l2lds.argtypes = [ctypes.c_void_p, ctypes.c_size_t] + [ctypes.c_int] * 7 + [ctypes.c_void_p, ctypes.c_void_p]
src = torch.randint(0, 255, (3 << 30,), dtype=torch.uint8, device="cuda")
sink = torch.zeros(4096, dtype=torch.int32, device="cuda")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
ITERS = 3072            # stages of 16 KB per workgroup


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
