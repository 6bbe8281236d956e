"""Child process of tests/test_gpu_multirank.py::test_trainer_n_shard_ranks_built_from_local_rows: one rank of a 2- or 8-rank (gloo,
all on cuda:0) data-parallel job.  The rank builds its model from ITS OWN rows of the data set (so models.py:18 gives it a local
num_data), wraps it in training.Trainer(group=...), evaluates one gradient on its slice of a job-wide injected noise draw and writes
the merged gradient to an .npz."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dgps_with_iwvi_amd import synthetic          # noqa: E402
from dgps_with_iwvi_amd.training import Trainer   # noqa: E402


def main():
    out, rows = sys.argv[1], [int(a) for a in sys.argv[2:]]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert len(rows) == world
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    spec = synthetic.make_spec(L=2, M=32, B=48, K=4, with_lv=True, seed=61, n_data=48)
    zs = synthetic.make_noise(spec, seed=62)
    lo = sum(rows[:rank])
    hi = lo + rows[rank]
    sub = dict(spec, X=spec["X"][lo:hi], Y=spec["Y"][lo:hi], B=hi - lo, n_data=hi - lo)
    model = synthetic.build_model(sub, dev)
    model.num_data = hi - lo                                    # what DGP_VI.__init__ sets for a rank built from its own rows
    tr = Trainer(model, group=dist.group.WORLD, shard="n", check_finite=False)
    assert model.num_data == sum(rows), model.num_data          # resolve_n_shard: the job's row count
    elbo, g = tr._gradients([torch.as_tensor(z[lo:hi], dtype=torch.float32, device=dev) for z in zs], advance=False)
    torch.cuda.synchronize()
    np.savez(out % rank, elbo=float(elbo), weight=tr.shard_weight, **{k: v.detach().double().cpu().numpy() for k, v in g.items()})
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
