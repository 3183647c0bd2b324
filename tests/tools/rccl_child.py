"""Child process of tests/test_gpu_rccl.py: a fresh interpreter that initialises torch.distributed with the
"nccl" backend (= RCCL on ROCm) in a world of ONE rank on cuda:0, sends the real weight blob through
``distributed.broadcast_weight_blob`` (forced: RCCL's broadcast kernel really runs), builds the device weights from
what arrived and checks one denoising step against weights that never went near a collective.  Exit code 0 = ok."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import torch.distributed as dist
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%s" % sys.argv[1], rank=0, world_size=1, device_id=dev)
    assert dist.get_backend() == "nccl"
    from diffusionsat_amd import distributed, runtime, weights as W
    from helpers import draws, make_all
    ws = W.glorot_weights(seed=7, bias_scale=0.05)
    host = W.to_blob(ws)
    blob = torch.from_numpy(host).to(dev)
    sent = blob.clone()
    distributed.broadcast_weight_blob(blob, src=0, force=True)          # ncclBroadcast on the RCCL communicator
    total = blob.double().sum()
    dist.all_reduce(total)                                              # and a reduction, for good measure
    torch.cuda.synchronize()
    assert torch.equal(blob, sent) and abs(total.item() - float(host.astype(np.float64).sum())) < 1e-6
    _, batch, _ = make_all("3sat")
    g = runtime.Graph(batch, device=dev)
    d = draws(batch.n_vars, 4, seed=2)
    noisy = torch.from_numpy(np.tile(np.array([[1, 0]], np.float32), (batch.n_vars, 1))).to(dev)
    kw = dict(normal=torch.from_numpy(d["normal"]).to(dev), labels=torch.from_numpy(d["labels"]).to(dev), rounds=4,
              early_exit=False, want_logits=True)
    a = runtime.Session(g, runtime.Weights(blob), max_rounds=4).diffusion_step(0.5, noisy, **kw)      # from the broadcast
    b = runtime.Session(g, runtime.Weights(ws), max_rounds=4).diffusion_step(0.5, noisy, **kw)        # from the host dict
    torch.cuda.synchronize()
    assert torch.equal(a["logits"], b["logits"]) and torch.isfinite(a["logits"]).all()
    merged = distributed.merge_counts({5: 2, 1 << 70: 1})
    assert merged == {5: 2, 1 << 70: 1}
    dist.destroy_process_group()
    print("RCCL_OK world=1 blob_floats=%d" % host.size)
    return 0


if __name__ == "__main__":
    sys.exit(main())
