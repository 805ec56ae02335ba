import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("speak-hack_amd"); ops = pkg.ops
from bench_encoder_layers import timeit
dev = torch.device("cuda:0"); B = 8
for (Cin, Cout, k, s, H) in [(64, 256, 1, 1, 64), (64, 64, 1, 1, 64), (256, 64, 1, 1, 64), (64, 64, 3, 1, 64), (3, 64, 7, 2, 256), (128, 512, 1, 1, 32)]:
    for G in (1, 6):
        Ho = ops.conv_out_size(H, k, s)
        first = Cin == 3
        x = torch.randn(B, Cin if first else G * Cin, H, H, device=dev)
        ws = [torch.randn(Cout, Cin, k, k, device=dev) * 0.05 for _ in range(G)]
        aff = (torch.rand(G * Cin, device=dev) + 0.5, torch.randn(G * Cin, device=dev) * 0.1)
        stats = torch.zeros(2 * G * Cout, device=dev, dtype=torch.float64)
        ncfg = pkg._lib.lib().spk_conv2d_num_configs()
        res = []
        for cfg in range(ncfg):
            if not ops.conv2d_config_fits(cfg, k, s, B, Cin, Cout, Ho, Ho):
                continue
            try:
                wp = torch.cat([ops.pack_conv_weight(w, cfg) for w in ws])
                r = []
                for a, st in ((None, None), (None, stats), (aff, None), (aff, stats)):
                    if first and a is not None:
                        r.append(0); continue
                    r.append(timeit(lambda: ops.conv2d_fused(x, wp, Cout, k, s, in_affine=a, stats=st, config=cfg, groups=G, shared_input=first)) * 1e3)
                res.append((cfg, ops.conv2d_config_info(cfg), r))
            except Exception as e:
                res.append((cfg, str(e)[:60], None))
        pick = ops.conv2d_pick_config(k, s, B, Cin, Cout, Ho, Ho)
        print(f"== {Cin}->{Cout} k{k} s{s} H{H} G{G} pick={pick}  [plain, stats, affine, both] us", flush=True)
        for cfg, info, r in res:
            print("   ", cfg, info, [round(v, 1) for v in r] if r else None, flush=True)
