"""CPU-only tests: the C ABI loads and exports every declared symbol, host-side logic (configs, gin
reader, schedule, replay buffer, flat gradient buffer, the data-parallel reduction over gloo with two
ranks) and the rule that the product never reaches the oracle.  No compute call touches a GPU here."""
import ctypes as C
import os
import re
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, make_dynamics, make_vhjb_config
from q_learning_with_hjb_amd import _abi


def test_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "hjbx.h")).read()
    body = hdr[hdr.index("#define HJBX_DECLARE"):hdr.index("HJBX_DECLARE(float, f32)")]
    typed = set(re.findall(r"\bint (hjbx_\w+)_##SFX\(", body))
    plain = set(re.findall(r"^(?:int|size_t|void) (hjbx_\w+)\(", hdr, flags=re.M))
    assert len(typed) == 13 and "hjbx_value_grad_f32" in plain and "hjbx_system_create" in plain
    declared = plain | {f"{t}_{s}" for t in typed for s in ("f32", "f64")}
    assert declared == set(_abi.EXPORTED_SYMBOLS), declared ^ set(_abi.EXPORTED_SYMBOLS)
    L = _abi.lib()
    for name in sorted(declared):
        assert hasattr(L, name), f"libhjbx.so does not export {name}"
    want = int(re.search(r"#define HJBX_VERSION (\d+)", open(os.path.join(ROOT, "include", "hjbx.h")).read()).group(1))
    assert L.hjbx_version() == want >= 100
    assert L.hjbx_reduce_workspace_bytes() >= 3 * 8


def test_descriptor_layouts_match_the_header():
    """ctypes structs == the C structs of include/hjbx.h (sizes checked against a gcc build of the header)."""
    from oracle import oracle as O
    O.lib()  # runs check_layout()
    assert C.sizeof(_abi.HjbxTask) == 8 * (100 + 9 + 9 + 100 + 10 + 3 + 10 + 10 + 1 + 1 + 1)


def test_system_create_errors_map_to_python_exceptions():
    with pytest.raises(ValueError, match="takes 4 parameters"):
        _abi.SystemHandle(_abi.SYS_CARTPOLE, 4, 1, 0.02, [-10], [10], [1, 0.1, 1])
    with pytest.raises(ValueError, match="unknown system kind"):
        _abi.SystemHandle(17, 4, 1, 0.02, [-10], [10], [1, 0.1, 1, 9.81])
    with pytest.raises(ValueError, match="dt must be positive"):
        _abi.SystemHandle(_abi.SYS_CARTPOLE, 4, 1, 0.0, [-10], [10], [1, 0.1, 1, 9.81])
    with pytest.raises(ValueError, match="umin"):
        _abi.SystemHandle(_abi.SYS_CARTPOLE, 4, 1, 0.02, [10], [-10], [1, 0.1, 1, 9.81])
    h = _abi.SystemHandle(_abi.SYS_NEARHOVER, 10, 3, 0.05, [0, -10, -10], [14.715, 10, 10], [9.81, 1, 0.91, 10])
    n, m = C.c_int(), C.c_int()
    assert _abi.lib().hjbx_dims(h.ptr, C.byref(n), C.byref(m)) == 0 and (n.value, m.value) == (10, 3)


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_no_cpu_fallback_without_a_gpu():
    """The hot path must fail loudly, never fall back to a CPU implementation."""
    d = make_dynamics("cartpole")
    assert d.get_dimension() == (4, 1) and d.dt == 0.02        # construction and metadata work anywhere
    with pytest.raises(RuntimeError, match="no HIP device"):
        d.simulate(np.zeros(4), np.zeros(1))
    with pytest.raises(RuntimeError, match="no HIP device"):
        d.get_initial_state()
    from q_learning_with_hjb_amd.controller.lqr import LQR
    lin = make_dynamics("linear")
    c = LQR(lin, np.eye(2), np.eye(1))                        # set-up math (CARE) is host side
    np.testing.assert_allclose(c.K, [[1.0, 3 ** 0.5]], rtol=1e-9)
    with pytest.raises(RuntimeError, match="no HIP device"):
        c.get_control_efforts(np.zeros(2))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "q_learning_with_hjb_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports the oracle"
                assert "liborc" not in src and "oracle_impl" not in src and "orc_" not in src, f"{f} references the oracle"


def test_configs_and_gin_lite(tmp_path):
    from q_learning_with_hjb_amd.configs import gin_lite
    from q_learning_with_hjb_amd.configs.controller.vhjb_controller_config import VHJBControllerConfig
    from q_learning_with_hjb_amd.configs.dynamics.dynamics_config import CartpoleDynamicsConfig
    cfg = make_vhjb_config("cartpole")
    assert cfg.Q.dtype == np.float32 and cfg.Q.shape == (4, 4) and cfg.features == [128, 128, 64]
    assert abs(float(cfg.xf[1]) - 3.1415926) < 1e-6 and cfg.maximum_step == 200 and cfg.batch_size == 256
    text = """
# comment
CartpoleDynamicsConfig.seed = 0
CartpoleDynamicsConfig.mc = 1
CartpoleDynamicsConfig.mp = 0.1
CartpoleDynamicsConfig.l = 1
CartpoleDynamicsConfig.g = 9.81
CartpoleDynamicsConfig.dt = 0.02
CartpoleDynamicsConfig.x0_mean = [0, 3.14, 0, 0]
CartpoleDynamicsConfig.x0_std = [2.4, 0.05,
                                 1, 0.05]
CartpoleDynamicsConfig.umin = [-10]
CartpoleDynamicsConfig.umax = [10]
"""
    f = tmp_path / "c.gin"
    f.write_text(text)
    gin_lite.clear_config()
    gin_lite.parse_config_file(str(f))
    c = CartpoleDynamicsConfig()
    assert c.state_dim == 4 and c.control_dim == 1 and c.x0_std.dtype == np.float32 and abs(c.mp - 0.1) < 1e-12
    c2 = CartpoleDynamicsConfig(mp=0.5)                       # explicit arguments win
    assert c2.mp == 0.5
    with pytest.raises(ValueError):
        gin_lite.parse_config("A.b = @other")
    with pytest.raises(TypeError):
        gin_lite.clear_config()
        VHJBControllerConfig()                                 # nothing bound -> the dataclass complains
    gin_lite.clear_config()


def test_sgdr_schedule_values():
    from q_learning_with_hjb_amd.controller.vhjb import sgdr_schedule
    kw = dict(init_value=0.0, peak_value=1e-5, end_value=0.0, warmup_steps=1000, decay_steps=2000, num_cycles=10)
    assert sgdr_schedule(0, **kw) == 0.0                                    # first update: zero termination weight (A.4)
    assert abs(sgdr_schedule(500, **kw) - 5e-6) < 1e-18
    assert abs(sgdr_schedule(1000, **kw) - 1e-5) < 1e-18
    assert abs(sgdr_schedule(1500, **kw) - 5e-6) < 1e-12
    assert abs(sgdr_schedule(2000, **kw)) < 1e-18 and abs(sgdr_schedule(2500, **kw) - 5e-6) < 1e-18
    assert sgdr_schedule(20000, **kw) < 1e-18 and sgdr_schedule(10 ** 6, **kw) < 1e-18


def test_replay_buffer_fifo_and_batches():
    from q_learning_with_hjb_amd.controller.vhjb import ReplayBuffer
    rb = ReplayBuffer(2, 10, torch.float32, "cpu")
    x = torch.arange(14, dtype=torch.float32)[:, None].repeat(1, 2)
    rb.extend(x[:6], x[:6, 0], torch.zeros(6))
    assert len(rb) == 6 and rb.num_batches(4) == 1
    rb.extend(x[6:14], x[6:14, 0], torch.ones(8))
    assert len(rb) == 10
    assert sorted(rb.x[:, 0].tolist()) == list(range(4, 14))          # FIFO: the 4 oldest records were evicted
    g = torch.Generator().manual_seed(0)
    seen = torch.cat([b[0][:, 0] for b in rb.batches(4, generator=g)])
    assert seen.numel() == 8 and len(set(seen.tolist())) == 8          # drop_last, without replacement
    assert list(rb.batches(16)) == []                                  # fewer records than a batch -> zero batches (A.4)
    rb.extend(x.repeat(3, 1), x[:, 0].repeat(3), torch.zeros(42))      # more than capacity at once keeps the newest
    assert len(rb) == 10 and sorted(rb.x[:, 0].tolist()) == list(range(4, 14))


def test_lecun_normal_statistics():
    from q_learning_with_hjb_amd.controller.vhjb import lecun_normal_
    w = lecun_normal_(torch.empty(128, 4096), torch.Generator().manual_seed(0))
    assert abs(float(w.std()) - (1 / 128) ** 0.5) < 2e-3 * (1 / 128) ** 0.5 * 10
    assert float(w.abs().max()) <= 2.0 * (1 / 128) ** 0.5 / 0.87962566103423978 + 1e-6


def test_pack_unpack_flat_roundtrip():
    from q_learning_with_hjb_amd.controller.vhjb import pack_flat, unpack_flat
    params = [torch.zeros(4, 128), torch.zeros(128, 128), torch.zeros(128, 64)]
    gh = [torch.randn_like(p) for p in params]
    gt = [torch.randn_like(p) for p in params]
    flat = pack_flat(gh, [gt[0], None, gt[2]], params, (torch.tensor(1.5), torch.tensor(2.5), torch.tensor(200.0), torch.tensor(56.0)))
    assert flat.numel() == 2 * 25088 + 4                               # 25,088 weights for n = 4
    a, b, sc = unpack_flat(flat, params)
    assert all(torch.equal(x, y) for x, y in zip(a, gh)) and torch.equal(b[0], gt[0]) and float(b[1].abs().max()) == 0
    assert [float(s) for s in sc] == [1.5, 2.5, 200.0, 56.0]


# ---- data-parallel reduction over gloo, world_size 2 ------------------------------------------------
def _shard_sums(xs, dones, costs, W):
    """A CPU stand-in for one rank's work, built from the ORACLE (test side only): loss sums for its shard
    of a cartpole minibatch as differentiable functions of the value-network weights."""
    from oracle import oracle as O
    d = make_dynamics("cartpole")
    cfg = make_vhjb_config("cartpole")
    s = O.System.from_dynamics(d)
    task = _abi.make_task(4, 1, cfg.Q, cfg.R, np.eye(4), cfg.xf, cfg.uf, cfg.obs_min, cfg.obs_max, cfg.epsilon)
    xf = torch.as_tensor(np.asarray(cfg.xf, np.float64))
    e = torch.as_tensor(O.wrap(s, (xs - xf).numpy()))
    a1 = e @ W[0]; a2 = torch.relu(a1) @ W[1]; y = torch.relu(a2) @ W[2]
    V = (y * y).sum(-1) + 1e-3 * (e * e).sum(-1)
    g = ((((2 * y) @ W[2].t()) * (a2 > 0)) @ W[1].t() * (a1 > 0)) @ W[0].t() + 2e-3 * e

    class Res(torch.autograd.Function):
        @staticmethod
        def forward(ctx, gg):
            li, dg, sums = O.hjb_residual(s, task, xs.numpy(), gg.detach().numpy(), dones.numpy())
            ctx.save_for_backward(torch.as_tensor(dg))
            return torch.tensor(sums[0])

        @staticmethod
        def backward(ctx, go):
            return go * ctx.saved_tensors[0]

    h_sum = Res.apply(g)
    t_sum = ((V / (costs + 1e-10) - 1).abs() * dones).sum()
    return h_sum, t_sum, (1 - dones).sum(), dones.sum()


def _dp_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
        from q_learning_with_hjb_amd.controller.vhjb import allreduce_and_mix
        data = torch.load(os.path.join(os.environ["HJBX_TEST_TMP"], "dp.pt"))
        W = [w.clone().requires_grad_(True) for w in data["W"]]
        lo, hi = data["splits"][rank], data["splits"][rank + 1]            # UNEVEN shards with different done counts
        hs, ts, ni, nd = _shard_sums(data["xs"][lo:hi], data["dones"][lo:hi], data["costs"][lo:hi], W)
        g_h = torch.autograd.grad(hs, W, retain_graph=True); g_t = torch.autograd.grad(ts, W)
        grads, hl, tl = allreduce_and_mix(g_h, g_t, (hs.detach(), ts.detach(), ni, nd), W, 0.25, 1e-10, None)
        if rank == 0:
            q.put(([g.numpy() for g in grads], float(hl), float(tl)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_data_parallel_reduction_equals_single_process(tmp_path, world):
    """G-rank result == 1-rank result on the same global minibatch (SURVEY 8e acceptance), gloo, world 2 / 4 / 8, UNEVEN shards.  From world 4
    on, one rank holds only interior samples (zero `done`: its termination sums and count are 0) and one rank only terminal ones (zero live
    samples: its hjb sums and count are 0) -- per-rank means would divide by zero / be wrong there; the flat buffer carries SUMS and counts
    and the division by the global counts happens after the ONE all-reduce (reference normalisers: controller/vhjb.py:241, 253)."""
    import torch.multiprocessing as mp
    from q_learning_with_hjb_amd.controller.vhjb import allreduce_and_mix
    torch.manual_seed(0)
    rng = np.random.default_rng(0)
    B = 96
    xs = torch.as_tensor(np.array([0, 3.1415926, 0, 0]) + rng.uniform(-1, 1, (B, 4)) * [1.0, 0.3, 1.0, 1.0])
    dones = torch.as_tensor((rng.uniform(size=B) < np.linspace(0.05, 0.7, B)).astype(np.float64))   # done density differs per shard
    costs = torch.as_tensor(rng.uniform(0.5, 9.0, B))
    cuts = {2: [0, 37, B], 4: [0, 30, 41, 70, B], 8: [0, 9, 20, 33, 41, 58, 70, 83, B]}[world]
    if world >= 4:
        dones[cuts[1]:cuts[2]] = 0.0                 # rank 1: no terminal sample at all
        dones[cuts[2]:cuts[3]] = 1.0                 # rank 2: no live sample at all
    W = [torch.randn(4, 16, dtype=torch.float64) * 0.5, torch.randn(16, 16, dtype=torch.float64) * 0.3, torch.randn(16, 8, dtype=torch.float64) * 0.3]
    torch.save(dict(xs=xs, dones=dones, costs=costs, W=W, splits=cuts), tmp_path / "dp.pt")
    os.environ["HJBX_TEST_TMP"] = str(tmp_path)
    # single process on the whole minibatch
    Ws = [w.clone().requires_grad_(True) for w in W]
    hs, ts, ni, nd = _shard_sums(xs, dones, costs, Ws)
    g_h = torch.autograd.grad(hs, Ws, retain_graph=True); g_t = torch.autograd.grad(ts, Ws)
    want, whl, wtl = allreduce_and_mix(g_h, g_t, (hs.detach(), ts.detach(), ni, nd), Ws, 0.25, 1e-10, False)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got, hl, tl = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert abs(hl - float(whl)) < 1e-12 and abs(tl - float(wtl)) < 1e-12
    for a, b in zip(got, want):
        np.testing.assert_allclose(a, b.numpy(), rtol=1e-10, atol=1e-13)


def test_inline_asm_lds_prefetch_is_register_safe(tmp_path):
    """The MFMA kernels prefetch LDS operands from inline asm and retire them with counted s_waitcnt (guide 5.7):
    the compiler must not touch a destination register between the asm load and its wait.  Audits the ISA of every
    instantiation (cross-compiles for gfx950, no GPU needed)."""
    import subprocess
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import audit_asm_loads
    outs, procs = [], []
    for act in (0, 1, 2, 3, 4):                               # one object per variant (relu, tanh, relu bf16x3, relu f16x2, sin), compiled side by side
        asm = tmp_path / f"mlp_act{act}.s"
        outs.append(asm)
        procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=on", "-S",
                                       "--cuda-device-only", f"-DHJBX_MLP_ACT={act}"] + (["-fno-slp-vectorize"] if act == 3 else []) +
                                      ["-o", str(asm), os.path.join(ROOT, "q_learning_with_hjb_amd", "csrc", "hjbx_mlp.hip")], stderr=subprocess.DEVNULL))
    train_asm = tmp_path / "train.s"
    procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=on", "-S", "--cuda-device-only",
                                   "-fno-slp-vectorize", "-o", str(train_asm), os.path.join(ROOT, "q_learning_with_hjb_amd", "csrc", "hjbx_train.hip")],
                                  stderr=subprocess.DEVNULL))
    coop_asm = tmp_path / "train_coop.s"
    procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=on", "-S", "--cuda-device-only",
                                   "-fno-slp-vectorize", "-o", str(coop_asm), os.path.join(ROOT, "q_learning_with_hjb_amd", "csrc", "hjbx_train_coop.hip")],
                                  stderr=subprocess.DEVNULL))
    for pr in procs:
        assert pr.wait() == 0
    for asm in outs + [train_asm, coop_asm]:
        assert audit_asm_loads.audit(str(asm)) == 0
        # the other direction (round 2's planar-quadrotor bug class): no inline-asm VALU statement reads an MFMA result still in flight
        assert audit_asm_loads.audit_mfma_asm_reads(str(asm)) == 0
    # ... and the audit does see such a read when there is one
    probe = tmp_path / "probe.s"
    probe.write_text("_Z5probev:\n\tv_mfma_f32_32x32x16_f16 v[64:79], v[0:3], v[4:7], v[64:79]\n\ts_nop 7\n\t;;#ASMSTART\n\tv_and_b32_e32 v30, v30, v64\n\t;;#ASMEND\n"
                     "\tv_mfma_f32_32x32x16_f16 v[80:95], v[0:3], v[4:7], v[80:95]\n\ts_nop 15\n\t;;#ASMSTART\n\tv_and_b32_e32 v31, v31, v80\n\t;;#ASMEND\n\ts_endpgm\n")
    assert audit_asm_loads.audit_mfma_asm_reads(str(probe)) == 1
    text = outs[0].read_text()
    assert text.count("v_mfma_f32_32x32x2_f32") > 10000 and "ds_read_b32" in text
    # No MFMA inference kernel of ANY variant (f32 relu, f32 tanh, bf16x3, f16x2, f32 sin) may fall back on scratch: most sit at 250-256 VGPRs, a
    # spilled value is reloaded behind an `s_waitcnt vmcnt(0)` that drains the log stores, and round 2 saw rollout kernels whose spill
    # store sat in an EXEC = 0 region return wrong trajectories (the cause -- a divergent `grp` -- is removed as well).
    import re
    meta = re.compile(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.sgpr_spill_count:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)")
    for asm, mfma, per in ((outs[0], "v_mfma_f32_32x32x2_f32", 700), (outs[1], "v_mfma_f32_32x32x2_f32", 700), (outs[2], "v_mfma_f32_32x32x16_bf16", 288),
                           (outs[3], "v_mfma_f32_32x32x16_f16", 288), (outs[4], "v_mfma_f32_32x32x2_f32", 700)):
        text = asm.read_text()
        assert text.count(mfma) >= 30 * per
        kernels = meta.findall(text)
        assert len(kernels) == 30 and sum("k_vhjb_rollout_mfma" in k[0] for k in kernels) == 23 and sum("k_value_grad_mfma" in k[0] for k in kernels) == 7
        for name, private, _sgpr_spill, vgpr_spill in kernels:
            assert int(private) == 0 and int(vgpr_spill) == 0, f"{name}: {private} bytes of scratch, {vgpr_spill} spilled VGPRs"
        if asm in (outs[2], outs[3]):
            assert "ds_read_b64_tr_b16" in text
        # `grp` is wave uniform to the compiler: taking the next tile group is a scalar branch, not an EXEC-masked region
    # the parameter-gradient chains (f32 and f16x2 instantiations): small stack objects are fine, REGISTER SPILLS are not
    text = train_asm.read_text()
    assert text.count("k_train_chains") > 36 and text.count("v_mfma_f32_32x32x16_f16") >= 18 * 384
    assert "Folded Spill" not in text and "Folded Reload" not in text
    for name, _private, _sgpr_spill, vgpr_spill in meta.findall(text):
        assert int(vgpr_spill) == 0, f"{name}: {vgpr_spill} spilled VGPRs"
    # the cooperative parameter-gradient kernel: 512 registers per wave (one wave per SIMD), its 192 outer-product accumulators in AGPRs,
    # and NO scratch (its first build hoisted ~200 loop-invariant LDS addresses out of the tile loop and spilled 78 of them)
    text = coop_asm.read_text()
    kernels = [k for k in meta.findall(text) if "k_train_coop" in k[0] and "reduce" not in k[0] and "update" not in k[0]]   # (not the two epilogue kernels)
    # 9 system instantiations x 2 residual modes x 2 activations (relu, tanh) x 2 tile splits (PS = 1, 4), + sin for the 6 systems with n <= 4
    assert len(kernels) == 9 * 4 * 2 + 6 * 2 * 2, len(kernels)
    for name, private, _sgpr_spill, vgpr_spill in kernels:
        assert int(private) == 0 and int(vgpr_spill) == 0, f"{name}: {private} bytes of scratch, {vgpr_spill} spilled VGPRs"
    assert text.count("v_mfma_f32_32x32x2_f32") >= 36 * 690 + 36 * 400


def test_graft_entry_build_check_passes():
    """The driver's "does it build" hook: compiles (no-op when current), loads the library, checks symbols and version."""
    import __graft_entry__ as g
    g.build()


def test_bench_self_launches_n_ranks():
    """`python bench.py --gpus 2` with no launcher must START two ranks (round 1 silently ran one): the --dry-run mode goes through
    the same self-launch path (fresh child processes, RANK / WORLD_SIZE / MASTER_* in their environment), rendezvous over gloo on the
    CPU, all-reduces the rank ids and prints the launch fields of the JSON line.  Also: a launcher world that contradicts --gpus is an
    error, not a silent single-rank run."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--scaling", "strong", "--steps", "7"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["rank_sum"] == 3 and line["ranks_seen"] == 2 and line["steps"] == 7 and line["scaling"] == "strong"
    assert line["shards"] == [[0, 1 << 19], [1 << 19, 1 << 20]] and line["global_batch"] == 1 << 20
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-run"], capture_output=True, text=True, timeout=120,
                       env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)


def test_oracle_passes_address_and_ub_sanitizers():
    """SURVEY section 5 (race / memory checking): the restatement the GPU kernels are judged against is itself run under
    AddressSanitizer + UBSan (`make -C oracle asan`: every batch entry point, both precisions, all five systems, ragged batch,
    exactly-sized heap buffers).  The GPU pool cannot run sanitizers, so this is the CPU-side check."""
    import subprocess
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-B", "asan"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "oracle sanitizer run: ok" in r.stdout


def test_value_network_arithmetic_option_roundtrip():
    """hjbx_set_option(HJBX_OPT_MLP_ARITHMETIC): the default is f32 (the reference's arithmetic; the 16-bit split modes are opt-in), query
    with a negative value, rejects unknown modes (host logic only)."""
    import q_learning_with_hjb_amd as pkg
    if not os.environ.get("HJBX_MLP_ARITHMETIC"):
        assert pkg.value_network_arithmetic() == "f32"
    prev = pkg.set_value_network_arithmetic("f32")
    try:
        assert pkg.set_value_network_arithmetic("f16x2") == "f32" and pkg.value_network_arithmetic() == "f16x2"
        assert pkg.set_value_network_arithmetic("bf16x3") == "f16x2"
        assert pkg.set_value_network_arithmetic("f32") == "bf16x3"
        with pytest.raises(ValueError):
            pkg.set_value_network_arithmetic("fp8")
        with pytest.raises(Exception):
            pkg._abi.set_option(pkg._abi.OPT_MLP_ARITHMETIC, 3)
        assert pkg.value_network_arithmetic() == "f32"
    finally:
        pkg.set_value_network_arithmetic(prev)


def test_value_network_arithmetic_from_the_environment():
    """HJBX_MLP_ARITHMETIC in the environment sets the option when the library is loaded (fresh interpreter)."""
    import subprocess
    code = "import q_learning_with_hjb_amd as p; print(p.value_network_arithmetic())"
    env = dict(os.environ, HJBX_MLP_ARITHMETIC="f16x2", PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().splitlines()[-1] == "f16x2", out.stderr[-400:]
    env.pop("HJBX_MLP_ARITHMETIC")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().splitlines()[-1] == "f32", out.stderr[-400:]       # the library default
    env["HJBX_MLP_ARITHMETIC"] = "fp8"
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "HJBX_MLP_ARITHMETIC" in out.stderr
