"""Exact GP regression on several MI355X: one covariance matrix factorised across all ranks (1-D block-cyclic columns,
every rank stores only its own block columns), predictions streamed from the distributed factor.  There is no reference
counterpart (GPflow-Slim is single-device); the model objects are the drop-in ones of examples/gpr.py.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 examples/gpr_distributed.py --npoints 131072
    ... --comm rccl      the collectives issued by the library's own RCCL binding (torch only carries the unique id)

Every rank builds the same model (same data, same hyper-parameters), calls the same functions and gets the same numbers
back, bit for bit.
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gpflow-slim_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--npoints", dest="n", type=int, default=32768)   # (not "--n": torch.distributed.run would take it for an abbreviation of its own options)
    ap.add_argument("--dims", dest="d", type=int, default=8)
    ap.add_argument("--num-test", dest="n_test", type=int, default=2048)
    ap.add_argument("--block", dest="nb", type=int, default=512, help="block-column width")
    ap.add_argument("--comm", default="torch", choices=["torch", "rccl"])
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for tests)")
    ap.add_argument("--force-device", type=int, default=-1, help="testing: put every rank on this GPU")
    args = ap.parse_args()

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    dev = args.force_device if args.force_device >= 0 else int(os.environ.get("LOCAL_RANK", "0"))
    os.environ["GPFLOWSLIM_DEVICE"] = str(dev)
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend)
    import gpflowSlim as gpf
    from gpflowSlim.distributed import RcclComm, SingleComm, TorchComm, gpr_lml_distributed, predict_f_distributed

    rng = np.random.default_rng(2024)                       # the same data on every rank
    X = rng.standard_normal((args.n, args.d))
    w = rng.standard_normal((args.d, 1)) / np.sqrt(args.d)
    f = lambda x: np.sin(x @ w)
    Y = f(X) + 0.1 * rng.standard_normal((args.n, 1))
    Xs = rng.standard_normal((args.n_test, args.d))

    kern = gpf.kernels.RBF(args.d, variance=1.0, lengthscales=np.sqrt(args.d) * np.ones(args.d), ARD=True)
    model = gpf.models.GPR(X, Y, kern, obs_var=0.01)
    if args.comm == "rccl":
        def carry(uid):
            box = [uid]
            if world > 1:
                dist.broadcast_object_list(box, src=0)
            return box[0]
        comm = RcclComm(gpf.get_handle(), rank, world, bootstrap=carry)
    else:
        comm = TorchComm() if world > 1 else SingleComm()

    t0 = time.perf_counter()
    lml = gpr_lml_distributed(model, comm, nb=args.nb, lookahead=2)          # partitioned storage: 8 N^2 / P bytes per rank
    t1 = time.perf_counter()
    mu, var = predict_f_distributed(model, Xs, comm)                          # streams the panels once more
    t2 = time.perf_counter()
    rmse = float(np.sqrt(np.mean((mu - f(Xs)) ** 2)))
    if rank == 0:
        print("ranks %d  N %d  log marginal likelihood %.6f  (%.3f s)  predict_f %d points (%.3f s)  rmse vs noise-free truth %.4f  "
              "device bytes on rank 0: %.2f GB" % (world, args.n, lml, t1 - t0, args.n_test, t2 - t1, rmse,
                                                    gpf.get_handle().device_bytes() / 1e9), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
