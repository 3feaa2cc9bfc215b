import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from kitti_motion_compensation_amd import capi
def run(label, pad=0):  # pad: empty frames appended so that the batch exceeds the kernel-argument tables (device tables: upload + host wait)
    ctx = capi.Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    for F, per in ((16, 1_000_000), (8, 1_000_000), (4, 1_000_000), (16, 123_397)):
        n = F * per
        R = max(2, int(600_000_000 // n) + 1)
        ins = [torch.empty((n, 4), dtype=torch.float32, device="cuda") for _ in range(R)]
        outs = [torch.empty_like(ins[0]) for _ in range(R)]
        for r in range(R): ctx.synth_points(ins[r], n, 7 + r)
        offs = np.concatenate([np.arange(F + 1, dtype=np.uint64) * per, np.full(pad, n, dtype=np.uint64)])
        prm = capi.params_array([capi.FrameParams.make([1.3, 0.02, 0, 0.001, -0.002, 0.03], 0.5)] * (F + pad))
        for k in range(40): ctx.deskew_batch_f32(ins[k % R], outs[k % R], offs, prm, None)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            ctx.timer_begin()
            for k in range(200): ctx.deskew_batch_f32(ins[k % R], outs[k % R], offs, prm, None)
            best = min(best, ctx.timer_end() / 200)
        print(f"{label}: {F:3d} x {per}: {best*1e3:8.2f} us per launch = {32*n/best/1e9:7.3f} TB/s", flush=True)
        del ins, outs
    ctx.close()
run("inline tables (kernel arguments)")
run("device tables (upload + host wait; 17 empty frames appended)", pad=17)
