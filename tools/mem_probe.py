import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from rspnet_amd.moco import ModelFactory, Loss
from rspnet_amd.optim import SGD
dev = torch.device("cuda", 0)
for arch, B, hw in (("c3d", 32, 112), ("resnet18", 32, 112), ("r2plus1d-vcop", 32, 112), ("s3dg", 16, 224)):
    torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
    cfg = {"model": {"arch": arch}, "moco": {"dim": 128, "k": 16384, "m": 0.999, "t": 0.07, "fc_type": "linear", "diff_speed": [2]}}
    model = ModelFactory(cfg).build_moco_diffloss(device=dev); model.train()
    crit = Loss(margin=2.0, A=1.0, M=1.0); opt = SGD(model.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-4)
    q = torch.randn(B, 3, 32, hw, hw, device=dev); k = torch.randn(B, 3, 32, hw, hw, device=dev)
    for _ in range(3):
        out, tgt, rl, rt = model(q, k); loss, _, _ = crit(out, tgt, rl, rt); opt.zero_grad(); loss.backward(); opt.step()
    torch.cuda.synchronize()
    print(f"{arch:14s} B={B} peak allocated {torch.cuda.max_memory_allocated()/2**30:6.2f} GiB  reserved {torch.cuda.max_memory_reserved()/2**30:6.2f} GiB", flush=True)
    del model, opt, q, k, out, loss
