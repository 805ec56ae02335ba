import sys, torch
r0 = torch.load(sys.argv[1] + "/rank0.pt", weights_only=False)
r1 = torch.load(sys.argv[1] + "/rank1.pt", weights_only=False)
for step in (0, 1):
    a, b = r0["steps"][step], r1["steps"][step]
    print("step", step, "loss", a["loss"], a["loss_local"], b["loss"], b["loss_local"], "norm", a["norm"], b["norm"])
    print(" by_hook", a["by_hook"], "by_finish", a["by_finish"], "n", a["n_buckets"], "cold", a["cold"])
    bad = 0
    for name, ga, gb, la, lb in zip(r0["names"], a["reduced"], b["reduced"], a["local"], b["local"]):
        if la is None:
            continue
        mean = (la.double() + lb.double()) / 2
        err = float((ga.double() - mean).norm() / mean.norm().clamp_min(1e-30))
        e0 = float((ga.double() - la.double()).norm() / la.double().norm().clamp_min(1e-30))
        e1 = float((ga.double() - lb.double()).norm() / lb.double().norm().clamp_min(1e-30))
        if err > 1e-6:
            bad += 1
            if bad < 12:
                print(f"  {name:40s} err {err:.3e} vs l0 {e0:.3e} vs l1 {e1:.3e} |ga| {float(ga.norm()):.3e} |mean| {float(mean.norm()):.3e} |la| {float(la.norm()):.3e} |lb| {float(lb.norm()):.3e} eq {torch.equal(ga, gb)}")
    print(" bad", bad, "of", len(r0["names"]))
