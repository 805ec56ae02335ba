import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("speak-hack_amd"); ops = pkg.ops
from bench_encoder_layers import timeit
dev = torch.device("cuda:0"); B = 8; G = 6
for (Cin, Cout, k, s, H) in [(3, 64, 7, 2, 256), (64, 256, 1, 1, 64), (64, 64, 3, 1, 64), (128, 512, 1, 1, 32)]:
    Ho = ops.conv_out_size(H, k, s)
    first = Cin == 3
    x = torch.randn(B, Cin if first else G * Cin, H, H, device=dev)
    ws = [torch.randn(Cout, Cin, k, k, device=dev) * 0.05 for _ in range(G)]
    cfg = ops.conv2d_pick_config(k, s, B, Cin, Cout, Ho, Ho)
    wp = torch.cat([ops.pack_conv_weight(w, cfg) for w in ws])
    r = [timeit(lambda: ops.conv2d_fused(x, wp, Cout, k, s, config=cfg, groups=G, shared_input=first)) * 1e3]
    for slots in (1, 4, 16, 32, 64, 128, 256, 1024):
        stats = torch.zeros(slots * 2 * G * Cout, device=dev, dtype=torch.float64)
        r.append(timeit(lambda: ops.conv2d_fused(x, wp, Cout, k, s, stats=stats, config=cfg, groups=G, shared_input=first)) * 1e3)
    print(f"{Cin}->{Cout} k{k} s{s} H{H} cfg{cfg} {ops.conv2d_config_info(cfg)}: plain, slots 1,4,16,32,64,128,256,1024:", [round(v, 1) for v in r], flush=True)
