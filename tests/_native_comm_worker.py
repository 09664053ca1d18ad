"""Worker of tests/test_gpu_comm_native.py::test_two_ranks_through_a_stand_in_transport: one rank of a two-process run of
the library's native communicator on ONE GPU, with tests/fake_rccl (a shared-memory stand-in for the RCCL transport) loaded
in place of librccl.  argv: rank world uid_file fake_lib out_file"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gpflow-slim_amd"), ROOT]
import numpy as np

rank, world, uid_file, fake, out_file = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
import gpflowSlim as gpf
from gpflowSlim import _backend as be
from gpflowSlim.distributed import RcclComm, gpr_lml_distributed, predict_f_distributed
from gpflowSlim.distributed_sparse import sparse_bound_distributed, conditional_distributed, svgp_bound_distributed
import oracle.gp_oracle as orc

be.comm_load(fake)                       # BEFORE anything asks for the real librccl
assert be.comm_version() == 29999        # the stand-in


def carry(uid):
    if rank == 0:
        with open(uid_file + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(uid_file + ".tmp", uid_file)
        return uid
    t0 = time.time()
    while not os.path.exists(uid_file):
        assert time.time() - t0 < 60
        time.sleep(0.05)
    return open(uid_file, "rb").read()


h = be.Handle(0)
be.set_handle(h)
comm = RcclComm(h, rank, world, bootstrap=carry)
res = {"rank": rank}
if os.environ.get("GPS_WORKER_MODE") == "fault":
    # tests/test_gpu_comm_native.py::test_a_failing_send_aborts_the_communicator_and_leaves_the_handle_usable: rank 1's transport
    # fails inside an open send / receive group (FAKE_RCCL_FAIL_*): every rank must come back with an error, without a
    # communicator, and with a handle that still evaluates
    n, d = 1500, 3
    X, Y, Xs = orc.synthetic_gpr_data(n, d, 10, seed=4)
    m = gpf.models.GPR(X, Y, gpf.kernels.RBF(d, variance=1.2, lengthscales=1.3), obs_var=0.1)
    try:
        gpr_lml_distributed(m, comm, nb=256, lookahead=2, partitioned=True)
        res["error"] = None
    except RuntimeError as e:
        res["error"] = str(e)
    try:
        h.comm_exchange(0, 16, 0, 0, 0)
        res["after"] = "exchange accepted"
    except RuntimeError as e:
        res["after"] = str(e)
    res["lml_after"] = m.compute_log_likelihood()
    spec = {"type": "rbf", "variance": orc.constrained(1.2), "lengthscales": orc.constrained(1.3), "input_dim": d}
    res["lml_ref"] = orc.gpr_lml(spec, X, Y, orc.constrained(0.1))
    comm.close()
    h.close()
    with open(out_file, "w") as f:
        json.dump(res, f)
    sys.exit(0)
n, d = 3000, 3
X, Y, Xs = orc.synthetic_gpr_data(n, d, 45, seed=13)
ls = np.linspace(0.9, 1.6, d)
m = gpf.models.GPR(X, Y, gpf.kernels.RBF(d, variance=1.2, lengthscales=ls, ARD=True), obs_var=0.1)
res["lml_part_sag"] = gpr_lml_distributed(m, comm, nb=256, lookahead=2, partitioned=True)          # scatter + all-gather
mu, var = predict_f_distributed(m, Xs, comm)
res["mu"], res["var"] = mu.tolist(), var.tolist()
comm.mode = "broadcast"
res["lml_repl_bcast"] = gpr_lml_distributed(m, comm, nb=512, lookahead=1, partitioned=False)
comm.mode = "scatter_allgather"
# the same through the Python schedule (RcclComm.exchange per panel) instead of gps_dist_lml / gps_dist_predict
comm.native_schedule = False
res["lml_part_py"] = gpr_lml_distributed(m, comm, nb=256, lookahead=2, partitioned=True)
mu2, var2 = predict_f_distributed(m, Xs, comm)
res["mu_py"] = mu2.tolist()
comm.native_schedule = True
res["exchanges"], res["bytes_sent"] = comm.exchanges, comm.bytes_sent
# config 5 pieces through the same communicator
rng = np.random.default_rng(5)
Z = X[:150].copy()
sg = gpf.models.SGPR(X, Y, gpf.kernels.RBF(d, variance=1.3, lengthscales=1.1), Z=Z, obs_var=0.15)
res["sgpr"] = sparse_bound_distributed(sg, comm, h)
fm, fv = conditional_distributed(Xs, Z, gpf.kernels.RBF(d, variance=1.3, lengthscales=1.1), np.sin(Z[:, :2]), comm=comm, handle=h)
res["cond_mean"] = fm.tolist()
sv = gpf.models.SVGP(X, Y, gpf.kernels.RBF(d, variance=1.3, lengthscales=1.1), gpf.likelihoods.Gaussian(0.2), Z=Z, q_diag=True)
res["svgp"] = svgp_bound_distributed(sv, comm, h)
comm.close()
h.close()
with open(out_file, "w") as f:
    json.dump(res, f)
