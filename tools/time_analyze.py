import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import fennec_amd
from fennec_amd import synth
ctx = fennec_amd.Context(0)
for name, mk in (("large_photo", lambda k: synth.large_photo(3840, 2160, k)), ("gradient", lambda k: synth.make_test_image(3840, 2160)),
                 ("solid", lambda k: synth.make_solid_image(3840, 2160, (90, 120, 33, 255)))):
    imgs = [torch.from_numpy(mk(k)).cuda() for k in range(16)]
    torch.cuda.synchronize()
    plan = ctx.plan_analyze_batch(imgs)
    ctx.profile(True)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        plan.run()
    ms = []
    for _ in range(20):
        plan.run(); ms.append(ctx.kernel_ms())
    print(f"{name:12s} analyze_pass_kernel {np.mean(ms) / 16 * 1e3:7.2f} us/img")
