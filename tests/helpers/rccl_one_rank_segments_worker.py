"""Child process of tests/test_gpu_multirank.py::test_graph_training_step_on_a_real_rccl_communicator: ONE rank on a real RCCL ("nccl")
communicator, the step's collectives forced on (sharding._FORCE_ONE_RANK_COLLECTIVES): what an 8-GPU training job captures into its step's
graph, minus the peers.  Eager steps and graph-segment steps from the same parameters and seed; prints one JSON line."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dgps_with_iwvi_amd import settings, sharding, synthetic   # noqa: E402
from dgps_with_iwvi_amd.training import Trainer                # noqa: E402


def main():
    import faulthandler
    faulthandler.dump_traceback_later(int(os.environ.get("IWVI_TEST_HANG_DUMP_S", "240")), exit=True)   # a hung collective ends the test with a traceback, not a timeout
    shard, steps = sys.argv[1], int(sys.argv[2])
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1)
    sharding._FORCE_ONE_RANK_COLLECTIVES = True
    spec = synthetic.make_spec(L=2, M=32, B=64, K=4, with_lv=True, seed=71, n_data=64)
    out = {}
    for mode in ("eager", "graph"):
        settings.set_seed(100)
        model = synthetic.build_model(spec, dev)
        tr = Trainer(model, group=dist.group.WORLD, shard=shard, use_graph=(mode == "graph"), check_finite=False, lr=1e-3, gamma=1e-3)
        vals = [float(tr.step()) for _ in range(steps)]
        torch.cuda.synchronize()
        out[mode] = {"elbo": vals, "params": [p.detach().double().cpu().numpy().tolist() for _, p, _ in tr._entries[:4]],
                     "q_mu": model.layers[-1].q_mu.detach().double().cpu().numpy().tolist()}
        if mode == "graph":
            g = tr._graphs["step"][1]
            out["n_graphs"], out["n_collectives"] = g.n_graphs, g.n_collectives
    out["same"] = bool(out["eager"]["elbo"] == out["graph"]["elbo"] and out["eager"]["params"] == out["graph"]["params"] and out["eager"]["q_mu"] == out["graph"]["q_mu"])
    out["finite"] = bool(np.all(np.isfinite(out["eager"]["elbo"])))
    print(json.dumps({k: out[k] for k in ("same", "finite", "n_graphs", "n_collectives")}))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
