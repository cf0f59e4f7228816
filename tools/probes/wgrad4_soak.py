"""Race screen of wgrad_mfma4_kernel: its ring protocol (counted vmcnt wait + one barrier per tick, slots recycled DEPTH + 3 ticks later)
must not depend on timing.  The same launch is repeated alone, beside a bandwidth-bound kernel on a second stream (memory latencies
stretch), and beside a second weight gradient (the CU's LDS / matrix pipe are shared), at three layer shapes and two workgroup counts;
every result must equal the first one bit for bit."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-mednet_amd")]
import torch
from mednet_hip import _lib as L

dev = "cuda:0"
lib = L.lib()
CL = torch.channels_last_3d
REPS = int(os.environ.get("SOAK_REPS", "60"))
bad = 0
for (n, cin, cout, s), wgs in [((4, 32, 32, 128), 0), ((4, 32, 32, 128), 128), ((4, 64, 64, 64), 0), ((2, 128, 96, (24, 40, 56)), 0), ((3, 32, 48, (33, 20, 70)), 24)]:
    d, h, w = (s, s, s) if isinstance(s, int) else s
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.randn(n, cin, d, h, w, device=dev, generator=g).bfloat16().contiguous(memory_format=CL)
    dy = torch.randn(n, cout, d, h, w, device=dev, generator=g).bfloat16().contiguous(memory_format=CL)
    big = torch.randn(64 * 1024 * 1024, device=dev)
    big2 = torch.empty_like(big)
    side = torch.cuda.Stream()
    main = torch.cuda.current_stream()

    def launch(stream, dw, ws):
        L.check(lib.mednet_conv3d_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), None, n, d, h, w, cin, cout, 3, L.BF16, L.NDHWC, L.BF16, L.NDHWC,
                                        L.ALGO_MFMA, wgs, ws.data_ptr(), ws.numel(), stream.cuda_stream), "wgrad")

    nbytes = lib.mednet_conv3d_wgrad_ws_bytes(n, d, h, w, cin, cout, 3, wgs)
    ws1, ws2 = (torch.empty(nbytes, dtype=torch.uint8, device=dev) for _ in range(2))
    ref = torch.empty(cout, cin, 3, 3, 3, device=dev)
    launch(main, ref, ws1)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(ref).all())
    diffs = 0
    for i in range(REPS):
        dw, dwb = torch.full_like(ref, float("nan")), torch.full_like(ref, float("nan"))
        ws1.fill_(0xFF)
        mode = i % 3
        side.wait_stream(main)
        if mode == 1:  # a streaming copy beside it
            with torch.cuda.stream(side):
                big2.copy_(big)
                big.copy_(big2)
        elif mode == 2:  # a second weight gradient beside it
            launch(side, dwb, ws2)
        launch(main, dw, ws1)
        main.wait_stream(side)
        torch.cuda.synchronize()
        if not torch.equal(dw, ref) or (mode == 2 and not torch.equal(dwb, ref)):
            diffs += 1
    print(f"wgrad {cin}->{cout} @{(d, h, w)} N={n} wgs={wgs or 256}: {REPS} launches (alone / beside a copy / beside itself), {diffs} differ from the first", flush=True)
    bad += diffs
print("RACE SCREEN", "FAILED" if bad else "clean")
sys.exit(1 if bad else 0)
