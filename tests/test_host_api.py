"""CPU tests of the host side: the gpflowSlim API mirror (no device work), the kernel-program
compiler, and the C-ABI library surface (loads, exports every declared symbol, refuses to run
without a GPU instead of falling back)."""
import ctypes
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gpf():
    lib = os.path.join(ROOT, "gpflow-slim_amd", "lib", "libgpflowslim_hip.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "gpflow-slim_amd", "csrc"), "-j4"])
    import gpflowSlim
    return gpflowSlim


def test_library_exports_every_declared_symbol(gpf):
    header = open(os.path.join(ROOT, "include", "gpflowslim_hip.h")).read()
    declared = set(re.findall(r"\b(gps_[a-z0-9_]+)\s*\(", header))
    declared -= {"gps_handle_s"}
    assert len(declared) >= 18
    lib = gpf.load_library()
    for name in sorted(declared):
        assert hasattr(lib, name), "libgpflowslim_hip.so does not export %s" % name
    from gpflowSlim import _backend
    assert declared == set(_backend.EXPORTED_SYMBOLS), "ctypes binding and header disagree"


def test_documented_options_are_the_accepted_options():
    """Every key gps_set_option accepts is documented in the public header's option list, and the list names no key the
    library does not take (option keys missing from the header; round 6 removed 25 keys with their designs)."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "gpflow-slim_amd", "csrc", "gps_handle.hip")).read()
    body = src[src.index('extern "C" int gps_set_option('):]
    body = body[:body.index("\n}\n")]
    accepted = set(re.findall(r'strcmp\(key, "([a-z_0-9]+)"\)', body))
    hdr = open(os.path.join(root, "include", "gpflowslim_hip.h")).read()
    block = hdr[hdr.index("/* options (diagnostics and A/B switches"):hdr.index("int gps_set_option(")]
    documented = set(re.findall(r'^ \*\s+(?:"[a-z_0-9]+"(?: / )?)+', block, flags=re.M) and re.findall(r'"([a-z_0-9]+)"', "\n".join(
        ln for ln in block.splitlines() if re.match(r'^ \*   "', ln))))
    assert len(accepted) <= 25, sorted(accepted)
    assert accepted <= documented, sorted(accepted - documented)
    stale = {k for k in documented - accepted if k not in ("lookahead_retries", "trsv_wave_fallbacks", "small_n_fallbacks", "small_n_cooldown")}
    assert not stale, sorted(stale)


def test_kern_node_struct_layout_matches_header(gpf):
    from gpflowSlim import _backend as be
    # int32 op, n_dims, active_dims[32]; double variance, period, lengthscales[32]
    assert ctypes.sizeof(be.KernNode) == 4 + 4 + 4 * 32 + 8 + 8 + 8 * 32
    assert be.KernNode.variance.offset == 136 and be.KernNode.lengthscales.offset == 152


def test_no_gpu_means_loud_failure_not_fallback(gpf):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        gpf.Handle(0)
    k = gpf.kernels.RBF(2)
    with pytest.raises(RuntimeError):
        k.K(np.zeros((3, 2)))
    m = gpf.models.GPR(np.zeros((3, 2)), np.zeros((3, 1)), k)
    with pytest.raises(RuntimeError):
        m.compute_log_likelihood()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "gpflow-slim_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "gp_oracle" not in txt and "import oracle" not in txt and "from oracle" not in txt, f


def test_parameter_and_transforms(gpf):
    P, T = gpf.params.Parameter, gpf.transforms
    p = P(2.5, transform=T.positive)
    assert float(p.value) == pytest.approx(2.5, rel=4e-16)
    assert float(p.unconstrained_tensor) == pytest.approx(2.5 - 1e-6 + np.log(-np.expm1(-(2.5 - 1e-6))), rel=1e-15)
    p.assign(0.1)
    assert float(p.value) == pytest.approx(0.1, rel=1e-15)
    p.assign_unconstrained(0.0)
    assert float(p.value) == pytest.approx(np.log(2.0) + 1e-6, rel=1e-15)       # softplus(0) + lower
    q = P(np.array([1.0, 2.0]), transform=T.Log1pe(1e-3))
    assert np.allclose(q.value, [1.0, 2.0], rtol=1e-15) and q.shape == (2,) and q.size == 2
    assert P(3.0).value == 3.0                                                    # Identity
    assert T.Log1pe().log_jacobian_tensor(np.array([0.0])) == pytest.approx(-np.log(2.0))
    assert gpf.settings.float_type is np.float64 and gpf.settings.numerics.jitter_level == 1e-6
    tmp = gpf.settings.get_settings(); tmp.numerics.jitter_level = 1e-3
    with gpf.settings.temp_settings(tmp):
        assert gpf.settings.jitter == 1e-3
    assert gpf.settings.jitter == 1e-6


def _ops(nodes):
    return [n.op for n in nodes]


def test_kernel_program_compilation(gpf):
    from gpflowSlim import _backend as be
    k = gpf.kernels
    rbf = k.RBF(3, variance=2.0, lengthscales=[1.0, 2.0, 3.0], ARD=True)
    (nd,) = rbf._nodes(False, 5)
    assert nd.op == be.K_RBF and nd.n_dims == 3 and list(nd.active_dims[:3]) == [0, 1, 2]
    assert nd.variance == pytest.approx(2.0, rel=1e-15) and list(nd.lengthscales[:3]) == pytest.approx([1, 2, 3], rel=1e-15)
    iso = k.Matern52(2, lengthscales=0.5, active_dims=[4, 1])
    (nd,) = iso._nodes(False, 5)
    assert nd.op == be.K_MATERN52 and list(nd.active_dims[:2]) == [4, 1] and list(nd.lengthscales[:2]) == pytest.approx([0.5, 0.5])
    (nd,) = iso._nodes(True, 2)                      # presliced: columns 0..input_dim-1
    assert list(nd.active_dims[:2]) == [0, 1]
    per = k.Periodic(3, period=2.0, lengthscales=1.5)
    (nd,) = per._nodes(False, 3)
    assert nd.op == be.K_PERIODIC and nd.period == pytest.approx(2.0) and nd.lengthscales[0] == pytest.approx(1.5)
    # left folds, flattening of same-class combinations, constants last (kernels.py:1019-1029,1073)
    s = rbf + iso + per + 2.0
    assert isinstance(s, k.Sum) and len(s.kern_list) == 3 and s.const_list == [2.0]
    assert _ops(s._nodes(False, 5)) == [be.K_RBF, be.K_MATERN52, be.K_ADD, be.K_PERIODIC, be.K_ADD, be.K_CONSTANT, be.K_ADD]
    p = (rbf + iso) * per * k.White(1, variance=0.1)
    assert isinstance(p, k.Product) and len(p.kern_list) == 3
    assert _ops(p._nodes(False, 5)) == [be.K_RBF, be.K_MATERN52, be.K_ADD, be.K_PERIODIC, be.K_MUL, be.K_WHITE, be.K_MUL]
    assert len(s.parameters) == 2 + 2 + 3
    X = np.zeros((4, 5))
    assert np.array_equal(s.Kdiag(X), np.full(4, float(rbf.variance) + float(iso.variance) + float(per.variance) + 2.0))
    assert np.allclose(p.Kdiag(X), (float(rbf.variance) + float(iso.variance)) * float(per.variance) * 0.1, rtol=1e-14)
    assert not s.on_separate_dimensions
    assert k.Sum([k.RBF(1, active_dims=[0]), k.RBF(1, active_dims=[1])]).on_separate_dimensions
    with pytest.raises(ValueError):
        be.make_program([be.op_node(be.K_ADD)] * 70)
    with pytest.raises(TypeError):
        k.Sum([rbf, "x"])


def test_gpr_shell_without_device(gpf):
    X = np.random.default_rng(0).standard_normal((5, 2)); Y = np.ones((5, 2))
    m = gpf.models.GPR(X, Y, gpf.kernels.RBF(2), obs_var=0.1)
    assert m.num_latent == 2 and len(m.parameters) == 3
    assert float(m.likelihood.variance) == pytest.approx(0.1, rel=1e-15)
    assert isinstance(m.mean_function, gpf.mean_functions.Zero) and m.mean_function(X).shape == (5, 1)
    assert m.prior_tensor == 0.0
    m2 = gpf.models.GPR(X, Y, gpf.kernels.RBF(2), obs_var=0.5, min_var=1e-2)
    assert float(m2.likelihood.variance) == pytest.approx(0.5, rel=1e-14)
    with pytest.raises(ValueError):
        gpf.models.GPR(X, np.ones(5), gpf.kernels.RBF(2))
    lin = gpf.mean_functions.Linear(np.ones((2, 2)), np.array([1.0, -1.0]))
    assert np.allclose(lin(X), X @ np.ones((2, 2)) + [1.0, -1.0])
    g = gpf.likelihoods.Gaussian(0.2)
    mu, var = g.predict_mean_and_var(np.zeros((3, 1)), np.ones((3, 1)))
    assert np.allclose(var, 1.2)
    assert np.allclose(g.predict_density(np.zeros((1, 1)), np.ones((1, 1)), np.zeros((1, 1))),
                       -0.5 * (np.log(2 * np.pi) + np.log(1.2)))
    feat = gpf.features.inducingpoint_wrapper(None, X)
    assert isinstance(feat, gpf.features.InducingPoints) and len(feat) == 5
    with pytest.raises(ValueError):
        gpf.features.inducingpoint_wrapper(feat, X)


def test_optimize_drivers_on_a_quadratic():
    """Model.optimize (models/model.py:172-196): the L-BFGS-B and Adam drivers, packing / unpacking of the
    unconstrained parameters and the trainable mask, on a model whose 'likelihood' is a concave quadratic
    (no GPU involved)."""
    import gpflowSlim as gpf
    from gpflowSlim.params import Parameter
    from gpflowSlim.models.model import Model

    class Quad(Model):
        def __init__(self):
            Model.__init__(self)
            self.a = Parameter(np.array([3.0, -2.0]), name="a")
            self.b = Parameter(0.5, name="b")
            self.c = Parameter(7.0, trainable=False, name="c")
            self._parameters = [self.a, self.b, self.c]
            self.target = [np.array([1.0, 2.0]), -1.5, 0.0]

        def _build_likelihood(self):
            return -sum(float(np.sum((np.atleast_1d(p.vf_val) - t) ** 2)) for p, t in zip(self._parameters, self.target))

        def compute_log_likelihood_and_gradients(self):
            return self._build_likelihood(), [(p, -2.0 * (p.vf_val - t)) for p, t in zip(self._parameters, self.target)]

    for method, kw in (("L-BFGS-B", {}), ("adam", {"learning_rate": 0.1, "max_iter": 600})):
        m = Quad()
        f = m.optimize(method=method, **kw)
        assert np.allclose(m.a.vf_val, [1.0, 2.0], atol=1e-3) and abs(float(m.b.vf_val) + 1.5) < 1e-3
        assert float(m.c.vf_val) == 7.0                       # not trainable: untouched
        assert f == pytest.approx(49.0, abs=1e-4)             # what the frozen parameter leaves
        assert m.objective == pytest.approx(f, abs=1e-9)


def test_bench_launcher_stays_off_the_gpu_and_reports_failure():
    """`python bench.py --gpus N` without a launcher starts the N ranks as fresh child processes; the parent must not
    import torch (nothing that could initialise a GPU) and must exit non-zero when a rank fails -- here every rank
    fails loudly because this machine has no GPU."""
    import subprocess
    code = ("import sys, os, tempfile; sys.path.insert(0, %r); import bench\n"
            "rc = bench.supervise([0, 1], 2, ['--gpus', '2', '--backend', 'gloo', '--steps', '1', '--warmup', '0'], ['torch'], tempfile.mkdtemp(), 120.0, 120.0, True)\n"
            "assert 'torch' not in sys.modules and 'gpflowSlim' not in sys.modules, sorted(sys.modules)\n"
            "print('launcher rc', rc)\n" % ROOT)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    import torch
    if torch.cuda.is_available():
        pytest.skip("the failure leg needs a machine without a GPU")
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "launcher rc 3" in p.stdout, (p.stdout, p.stderr[-2000:])
    assert "needs an MI355X" in p.stderr and "exited with status" in p.stderr
    # and through the command line: same thing, non-zero status -- and the one JSON line rank 0 leaves says why (value 0)
    q = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo"], capture_output=True,
                       text=True, timeout=300, env=env)
    lines = [json.loads(ln) for ln in q.stdout.splitlines() if ln.startswith("{")]
    assert q.returncode != 0 and len(lines) == 1 and lines[0]["value"] == 0.0 and "needs an MI355X" in lines[0]["error"]
    assert lines[0]["launch"]["tried"] == ["torch"] and len(lines[0]["launch"]["failed_attempts"]) == 1


def test_bench_supervisor_walks_through_its_attempts():
    """Every attempt of a multi-rank run is a set of FRESH rank processes started by the GPU-free supervisor; a failed attempt
    is followed by the next one, and the one JSON line says what was tried and why it failed.  Here (no GPU) every attempt
    fails: both forms -- bench.py as its own launcher, and one bench.py per rank under a launcher, whose per-rank supervisors
    meet in a rendezvous directory -- must walk through the whole list and end with ONE line from rank 0."""
    import tempfile
    import torch
    if torch.cuda.is_available():
        pytest.skip("the failure legs need a machine without a GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    args = ["--gpus", "2", "--backend", "gloo", "--attempts", "rccl,torch,torch-broadcast", "--steps", "1", "--warmup", "0"]
    q = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=300, env=env)
    lines = [json.loads(ln) for ln in q.stdout.splitlines() if ln.startswith("{")]
    assert q.returncode != 0 and len(lines) == 1, (q.stdout, q.stderr[-2000:])
    la = lines[0]["launch"]
    assert la["tried"] == ["rccl", "torch", "torch-broadcast"] and [f["attempt"] for f in la["failed_attempts"]] == la["tried"]
    assert all("exited with status" in f["reason"] for f in la["failed_attempts"])
    # one supervisor per rank (what a launcher such as torch.distributed.run starts)
    rdv = tempfile.mkdtemp(prefix="gps_bench_test_")
    procs = []
    for r in (0, 1):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999", GPS_BENCH_RDV_DIR=rdv)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py")] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                      text=True, env=e))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode != 0 for p in procs)
    lines0 = [json.loads(ln) for ln in outs[0][0].splitlines() if ln.startswith("{")]
    assert len(lines0) == 1 and not [ln for ln in outs[1][0].splitlines() if ln.startswith("{")]
    assert [f["attempt"] for f in lines0[0]["launch"]["failed_attempts"]] == ["rccl", "torch", "torch-broadcast"]


def test_native_rccl_binding_loads_without_a_gpu(gpf):
    """csrc/comm_rccl.hip binds librccl with dlopen on first use: the library itself still links libamdhip64 only, RCCL
    resolves (version, a unique id) on a machine without a GPU, and a communicator cannot be had without one."""
    from gpflowSlim import _backend as be
    lib = os.path.join(ROOT, "gpflow-slim_amd", "lib", "libgpflowslim_hip.so")
    needed = subprocess.run(["readelf", "-d", lib], capture_output=True, text=True).stdout
    assert "rccl" not in needed.lower() and "libamdhip64" in needed
    be.comm_load()
    assert be.comm_version() >= 20000
    uid = be.comm_unique_id()
    assert isinstance(uid, bytes) and len(uid) == 128 and uid != be.comm_unique_id()


def test_load_of_a_missing_library_fails_cleanly():
    """gps_comm_load of a path that does not exist must come back with an error (and fall through to the system's librccl),
    not crash on a NULL dlerror() string."""
    import subprocess
    import sys
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path[:0] = [%r]; from gpflowSlim import _backend as be\n"
            "be.comm_load('/nonexistent/librccl_nowhere.so'); print('version', be.comm_version())\n" % os.path.join(root, "gpflow-slim_amd"))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and "version" in p.stdout, (p.stdout, p.stderr[-2000:])


def test_bench_launcher_takes_its_ranks_with_it():
    """Whoever stops `python bench.py --gpus N` (a driver's time-out: SIGTERM) stops the rank processes it started too --
    no orphaned GPU processes."""
    import signal
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo"], stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, env=env)
    kids = []
    for _ in range(50):
        time.sleep(0.1)
        kids = subprocess.run(["pgrep", "-P", str(p.pid)], capture_output=True, text=True).stdout.split()
        if len(kids) == 2:
            break
    assert len(kids) == 2, kids
    p.send_signal(signal.SIGTERM)
    p.wait(timeout=30)
    time.sleep(0.5)

    def alive(pid):
        try:
            return "zombie" not in open("/proc/%s/status" % pid).read().lower()
        except OSError:
            return False
    assert p.returncode == 128 + signal.SIGTERM
    assert not [k for k in kids if alive(k)]
