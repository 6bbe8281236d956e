"""Child process of tests/test_gpu_multirank.py::test_sharded_training_step_as_graph_segments: one rank of a 2-rank (gloo, both on cuda:0)
training job, N- or K-sharded.  The rank runs five steps with the EAGER sharded trainer and five with Trainer(use_graph=True) -- the step
captured as hipGraph segments with the job's collectives between them -- from the same parameters and noise seed, and writes both
parameter sets, the bounds seen, the segment / collective counts and the per-step times to an .npz."""
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dgps_with_iwvi_amd import settings, synthetic   # noqa: E402
from dgps_with_iwvi_amd.training import Trainer      # noqa: E402


def main():
    import faulthandler
    faulthandler.dump_traceback_later(int(os.environ.get("IWVI_TEST_HANG_DUMP_S", "240")), exit=True)   # a hung collective ends the test with a traceback, not a timeout
    out, shard, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    B, K = 64, 4
    spec = synthetic.make_spec(L=2, M=32, B=B * (world if shard == "n" else 1), K=K, with_lv=True, seed=71, n_data=B * world)
    if shard == "n":                                             # this rank's rows of the job's minibatch
        lo, hi = rank * B, (rank + 1) * B
        spec = dict(spec, X=spec["X"][lo:hi], Y=spec["Y"][lo:hi], B=B, n_data=B)
    res = {}
    for mode in ("eager", "graph"):
        settings.set_seed(100 + rank)                            # (every rank its own noise; the same for both modes)
        model = synthetic.build_model(spec, dev)
        if shard == "n":
            model.num_data = B
        tr = Trainer(model, group=dist.group.WORLD, shard=shard, use_graph=(mode == "graph"), check_finite=False, lr=1e-3, gamma=1e-3)
        vals = []
        for _ in range(steps):
            vals.append(float(tr.step()))
        torch.cuda.synchronize()
        res[mode + ".elbo"] = np.asarray(vals)
        for name, p, _ in tr._entries:
            res[mode + "." + name] = p.detach().double().cpu().numpy()
        res[mode + ".q_mu"] = model.layers[-1].q_mu.detach().double().cpu().numpy()
        res[mode + ".q_sqrt"] = model.layers[-1].q_sqrt.detach().double().cpu().numpy()
        # per-step time (both ranks in step: a barrier on both sides)
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            tr.step()
        torch.cuda.synchronize()
        dist.barrier()
        res[mode + ".ms"] = (time.perf_counter() - t0) / 10 * 1e3
        if mode == "graph":
            g = tr._graphs["step"][1]
            res["n_graphs"], res["n_collectives"] = g.n_graphs, g.n_collectives
    np.savez(out % rank, **res)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
