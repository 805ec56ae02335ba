import importlib, sys, os, torch
sys.path.insert(0, os.getcwd())
pkg = importlib.import_module("speak-hack_amd"); ops = pkg.ops
sg2 = importlib.import_module("speak-hack_amd.stylegan2")
k = sg2.make_kernel((1,3,3,1))
for (shape, up, down, pad) in [((8,3,128,128),2,1,(2,1)), ((8,128,128,128),2,1,(2,1)), ((8,128,256,256),1,2,(1,1)), ((8,64,256,256),1,1,(2,1))]:
    x = torch.randn(*shape, device="cuda")
    kk = k * up * up
    for _ in range(3): y = ops.upfirdn2d(x, kk, up, down, pad)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): y = ops.upfirdn2d(x, kk, up, down, pad)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    gb = (x.numel() + y.numel()) * 4 / 1e9
    print(shape, up, down, f"{ms*1e3:.1f} us  {gb/ms*1e3:.0f} GB/s")
