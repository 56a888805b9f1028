"""Depth camera of the vision student: qa_tsc_depth_update = update_depth_buffer + process_depth_image
(tsc/legged_gym/envs/base/legged_robot.py:154-200) with the camera of attach_camera (:1203-1226), ray-cast against the course's
height field + ceiling field.  The reference's images come out of Isaac Gym's rasteriser (absent here): the camera model and the
crop / clip / normalise / noise / ring arithmetic are pinned by analytic scenes (CPU, oracle twin) and the HIP kernel against the
oracle on a real course (GPU)."""
import ctypes as C

import numpy as np
import pytest
import torch

from quadrupedal_agility_amd import _capi
from quadrupedal_agility_amd.tsc.legged_gym.task_level import coarse_depth_maps
from tests.oracle_lib import load_oracle

W, H, HC, WC = 106, 60, 58, 87
NEAR, FAR = 0.3, 4.0
TAN_H = np.tan(np.radians(87.0) / 2); TAN_V = TAN_H * H / W
CAM = np.array([0.305, 0.0175, 0.098])


def depth_cfg(n, rows, cols, noise=0.0, step=0, seed=1, buffer_len=2, border=1.0, hs=0.05, vs=0.005):
    c = _capi.QaTscDepthCfg()
    c.num_envs, c.step, c.seed, c.env_id_offset = n, step, seed, 0
    c.width, c.height, c.crop_top, c.crop_bottom, c.crop_left, c.crop_right = W, H, 1, 1, 10, 9
    c.buffer_len, c.map_rows, c.map_cols = buffer_len, rows, cols
    c.horizontal_fov_deg = 87.0
    for i in range(3):
        c.position[i] = CAM[i]
    c.near_clip, c.far_clip, c.depth_noise = NEAR, FAR, noise
    c.border_size, c.horizontal_scale, c.vertical_scale = border, hs, vs
    return c


def render(fn, cfg, root, pitch, hmap, cmap, ep_len, buf, stream=None, coarse=None):
    io = _capi.QaTscDepthIo()
    keep = [root, pitch, hmap, cmap, ep_len, buf, coarse]
    if coarse is not None:
        cfg.coarse_log2 = 3
        io.coarse_floor_max = coarse[0].data_ptr()
        io.coarse_ceiling_min = coarse[1].data_ptr() if coarse[1] is not None else None
    io.root_states, io.camera_pitch, io.height_samples = root.data_ptr(), pitch.data_ptr(), hmap.data_ptr()
    io.ceiling_samples = cmap.data_ptr() if cmap is not None else None
    io.episode_length, io.depth_buffer = ep_len.data_ptr(), buf.data_ptr()
    rc = fn(C.byref(cfg), C.byref(io), stream)
    assert rc == 0, rc
    return keep


def oracle_fn():
    lib = load_oracle()
    f = lib.qo_tsc_depth_update; f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]; f.restype = C.c_int
    return f


def scene(n=1, rows=200, cols=200, z=0.35, yaw=0.0):
    root = torch.zeros(n, 13); root[:, 0] = 2.0; root[:, 1] = 4.0; root[:, 2] = z
    root[:, 5] = np.sin(yaw / 2); root[:, 6] = np.cos(yaw / 2)
    return root, torch.zeros(n), torch.zeros(rows, cols, dtype=torch.int16), torch.ones(n, dtype=torch.long) * 5, torch.zeros(n, 2, HC, WC)


def pixel_rays(pitch=0.0):
    """camera-frame ray of every cropped pixel, rotated into the (yaw-free) trunk frame"""
    r = np.arange(HC)[:, None] + 1; c = np.arange(WC)[None, :] + 10
    sx = ((c + 0.5) / W * 2 - 1) * TAN_H; sy = ((r + 0.5) / H * 2 - 1) * TAN_V
    ca, sa = np.cos(pitch), np.sin(pitch)
    return np.broadcast_arrays(ca - sy * sa, -sx + 0 * sy, -sa - sy * ca)


def norm(d):
    return (np.clip(d, NEAR, FAR) - NEAR) / (FAR - NEAR) - 0.5


@pytest.mark.parametrize("pitch_deg", [0.0, 5.0, -5.0])
def test_flat_floor_depth_is_camera_height_over_ray_slope(pitch_deg):
    """a plane is hit at t = h_cam / (-d_z): exact for the march's linear interpolation; rays at or above the horizon read `far`"""
    root, pitch, hmap, ep, buf = scene()
    pitch[:] = np.radians(pitch_deg)
    render(oracle_fn(), depth_cfg(1, 200, 200), root, pitch, hmap, None, ep, buf)
    dx, dy, dz = pixel_rays(np.radians(pitch_deg))
    hc = 0.35 + CAM[2]
    with np.errstate(divide="ignore"):
        t = np.where(dz < 0, hc / -dz, np.inf)
    img = buf[0, -1].numpy()
    assert np.abs(img - norm(t)).max() < 2e-6
    assert (img[:20] == 0.5).all() and img[-1, WC // 2] < -0.2          # sky rows = far; the bottom row sees the floor ~1 m ahead


@pytest.mark.parametrize("yaw", [0.0, np.pi / 2, -2.0])
def test_wall_depth_is_the_distance_along_the_optical_axis(yaw):
    """a 2 m wall 1.5 m ahead of the trunk origin (whatever the heading): every pixel whose ray reaches the wall before the floor
    reads the PLANAR depth wall - camera_x (a depth image, not a range image), to within the one-cell ramp of the height field"""
    root, pitch, hmap, ep, buf = scene(yaw=yaw)
    xs = (np.arange(200) * 0.05 - 1.0)[:, None] + 0 * np.arange(200)[None, :]; ys = xs.T
    ahead = (xs - 2.0) * np.cos(yaw) + (ys - 4.0) * np.sin(yaw)          # distance ahead of the trunk origin
    hmap[torch.from_numpy(ahead >= 1.5)] = 400
    render(oracle_fn(), depth_cfg(1, 200, 200), root, pitch, hmap, None, ep, buf)
    img = buf[0, -1].numpy()
    dx, dy, dz = pixel_rays()
    wall_t = (1.5 - CAM[0])
    hits_wall = (0.35 + CAM[2] + wall_t * dz > 0.05) & (0.35 + CAM[2] + wall_t * dz < 1.9)
    if yaw == 0.0:                                                        # axis-aligned: the ramp is one cell wide
        assert np.abs(img[hits_wall] - norm(wall_t)).max() < 0.06 / (FAR - NEAR)
    else:                                                                 # a staircase of cells: within one cell diagonal either way
        assert np.abs(img[hits_wall] - norm(wall_t)).max() < 0.15 / (FAR - NEAR)
    assert hits_wall.mean() > 0.5


def test_ceiling_is_seen_from_below_and_missing_triangles_are_not():
    root, pitch, hmap, ep, buf = scene()
    cmap = torch.full((200, 200), 32767, dtype=torch.int16)
    cmap[:, :] = 160                                                       # 0.8 m roof everywhere...
    cmap[:, 100:] = 32767                                                  # ... except over y >= 4.0 (map col 100 = y 4.0)
    render(oracle_fn(), depth_cfg(1, 200, 200), root, pitch, hmap, cmap, ep, buf)
    img = buf[0, -1].numpy()
    dx, dy, dz = pixel_rays()
    hc = 0.35 + CAM[2]
    t_roof = np.where(dz > 0, (0.8 - hc) / np.maximum(dz, 1e-9), np.inf)
    y_hit = 4.0 + CAM[1] + t_roof * dy
    roofed = (dz > 0.05) & (y_hit < 3.9)                                   # rays that reach the roof where it exists (image right half: -y)
    open_sky = (dz > 0.05) & (4.0 + CAM[1] + FAR * dy > 4.1) & (4.0 + CAM[1] + 0.0 * dy + dy * 0 > 0) & (dy > 0.02)
    assert np.abs(img[roofed] - norm(t_roof[roofed])).max() < 2e-6
    assert (img[open_sky] == 0.5).all() and roofed.sum() > 500 and open_sky.sum() > 500


def test_ring_and_noise():
    """ring (:196-200): all slots = the image where episode_length <= 1, else shift; noise (:166-168): one offset + one amplitude per
    image, U(-1,1) per pixel, keyed by (seed; env, step)"""
    f = oracle_fn()
    root, pitch, hmap, ep, buf = scene(n=2)
    ep[0] = 1; ep[1] = 7
    buf[:] = torch.arange(2 * 2 * HC * WC, dtype=torch.float32).view(2, 2, HC, WC)
    old = buf.clone()
    render(f, depth_cfg(2, 200, 200), root, pitch, hmap, None, ep, buf)
    clean = buf[0, 1].clone()
    assert torch.equal(buf[0, 0], buf[0, 1]) and torch.equal(buf[1, 0], old[1, 1]) and torch.equal(buf[1, 1], clean)
    out = []
    for step in (3, 3, 4):
        b = torch.zeros(2, 2, HC, WC)
        render(f, depth_cfg(2, 200, 200, noise=0.05, step=step), root, pitch, hmap, None, ep, b)
        out.append(b[:, 1] - clean)
    assert torch.equal(out[0], out[1]) and not torch.equal(out[0], out[2])                # a function of the key only
    assert not torch.equal(out[0][0], out[0][1])                                            # envs draw independently
    for e in range(2):
        n = out[0][e].numpy().astype(np.float64)
        offs = 0.5 * (n.max() + n.min())                                                   # offset + amp * U(-1, 1)
        amp = 0.5 * (n.max() - n.min())
        assert abs(offs) <= 0.05 + 1e-6 and 0 <= amp <= 0.05 + 1e-6
        u = (n - offs) / amp
        assert abs(u.mean()) < 0.05 and abs(u.std() - 1 / np.sqrt(3)) < 0.02               # uniform on (-1, 1)


def test_coarse_maps_bound_every_sample_their_cells_touch():
    rng = np.random.default_rng(0)
    h = torch.from_numpy(rng.integers(-50, 400, (483, 601)).astype(np.int16))
    c = torch.from_numpy(np.where(rng.random((483, 601)) < 0.3, rng.integers(80, 300, (483, 601)), 32767).astype(np.int16))
    fmax, cmin = coarse_depth_maps(h, c, 3)
    assert fmax.shape == cmin.shape == (((483 - 2) >> 3) + 1, ((601 - 2) >> 3) + 1)
    for (i, j) in [(0, 0), (5, 7), (fmax.shape[0] - 1, fmax.shape[1] - 1), (fmax.shape[0] - 1, 3)]:
        blk = (slice(8 * i, 8 * i + 9), slice(8 * j, 8 * j + 9))
        assert fmax[i, j] == h[blk].max() and cmin[i, j] == c[blk].min()


# ------------------------------------------------------------------ HIP vs oracle
def course_scene(n, seed):
    from quadrupedal_agility_amd.tsc.legged_gym.envs.base.legged_robot_config import LeggedRobotCfg
    from quadrupedal_agility_amd.tsc.legged_gym.utils.obstacle import Obstacle
    ob = Obstacle(LeggedRobotCfg.obstacle(), n, seed=seed)
    rng = np.random.default_rng(seed)
    root = torch.zeros(n, 13)
    goals = ob.flat_goals()                                                 # robots stand near goals, facing anywhere
    pick = rng.integers(0, goals.shape[1], n)
    root[:, :2] = torch.from_numpy(goals[np.arange(n), pick, :2] + rng.uniform(-0.4, 0.4, (n, 2))).float()
    ix = ((root[:, 0].numpy() + 5.0) / 0.05).astype(int); iy = ((root[:, 1].numpy() + 5.0) / 0.05).astype(int)
    root[:, 2] = torch.from_numpy(ob.height_field_raw[ix, iy] * 0.005 + rng.uniform(0.25, 0.45, n)).float()
    rpy = np.stack([rng.uniform(-0.3, 0.3, n), rng.uniform(-0.4, 0.4, n), rng.uniform(-np.pi, np.pi, n)], 1)
    cr, sr, cp, sp, cy, sy = (f(rpy[:, i] / 2) for i in range(3) for f in (np.cos, np.sin))
    root[:, 3:7] = torch.from_numpy(np.stack([sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy,
                                              cr * cp * cy + sr * sp * sy], 1)).float()
    pitch = torch.from_numpy(np.radians(rng.uniform(-5, 5, n))).float()
    ep = torch.from_numpy(rng.integers(0, 4, n)).long()
    return ob, root, pitch, ep


@pytest.mark.gpu
@pytest.mark.parametrize("n,seed", [(16, 1), (64, 7), (512, 11)])       # 512 = one GPU's share of BASELINE configs[4] (4096 camera envs over 8 GPUs)
def test_hip_depth_matches_oracle_on_a_course(n, seed):
    """same cameras, same course: the fp32 kernel and the fp64 oracle agree to 1e-4 of the normalised range on all but the pixels
    whose ray grazes a silhouette edge (a different march step wins); noise and ring are bit-identical functions of the key"""
    ob, root, pitch, ep = course_scene(n, seed)
    hmap = torch.from_numpy(np.ascontiguousarray(ob.height_field_raw)); cmap = torch.from_numpy(np.ascontiguousarray(ob.ceiling_raw))
    rows, cols = hmap.shape
    lib = _capi.load_library()
    start = torch.randn(n, 2, HC, WC)
    res = {}
    for noise in (0.0, 0.05):
        cfg = depth_cfg(n, rows, cols, noise=noise, step=11, seed=3, border=5.0)
        bo = start.clone(); render(oracle_fn(), cfg, root, pitch, hmap, cmap, ep, bo)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        bh, bh_full = start.clone().cuda(), start.clone().cuda()
        render(lib.qa_tsc_depth_update, cfg, root.cuda(), pitch.cuda(), hmap.cuda(), cmap.cuda(), ep.cuda(), bh, st,
               coarse=coarse_depth_maps(hmap.cuda(), cmap.cuda(), 3))
        render(lib.qa_tsc_depth_update, depth_cfg(n, rows, cols, noise=noise, step=11, seed=3, border=5.0), root.cuda(), pitch.cuda(), hmap.cuda(),
               cmap.cuda(), ep.cuda(), bh_full, st)
        torch.cuda.synchronize()
        assert torch.equal(bh, bh_full)                       # the empty-space skipping changes no pixel
        d = (bh.cpu() - bo).abs()
        res[noise] = d
        frac = (d > 1e-4).float().mean().item()
        print(f"noise {noise}: pixels off by > 1e-4: {frac:.4%}, median {d.median().item():.2e}, seen-something fraction {(bo[:, 1].abs() < 0.499).float().mean().item():.2f}")
        assert frac < 0.01
        assert torch.equal((bh.cpu()[:, 0] == bh.cpu()[:, 1]).flatten(1).all(1), ep <= 1)
    assert (res[0.05] > 1e-4).float().mean() <= (res[0.0] > 1e-4).float().mean() + 1e-4      # the noise adds no disagreement
    assert (bo[:, 1] - 0.5).abs().max() > 0.5                                                 # cameras see obstacles / floor, not only sky


# ------------------------------------------------------------------ the env's depth buffer (CPU: oracle twins under the same host code)
def test_env_hands_the_previous_image_to_the_runner():
    """extras["depth"] = depth_buffer[:, -2] (:145-146): the image rendered one env step earlier; the ring restarts with the episode;
    the camera pitch is a per-env constant inside depth.angle"""
    from tests.test_tsc_course_env import cpu_env
    env = cpu_env(4, seed=5, depth__use_camera=True)
    assert env.depth_buffer.shape == (4, 2, HC, WC)
    pitch = np.degrees(env.bk.camera_pitch.numpy())
    assert (pitch >= -5).all() and (pitch <= 5).all() and np.unique(pitch).size == 4
    assert torch.equal(env.depth_buffer[:, 0], env.depth_buffer[:, 1])              # first frame of the episode fills the ring
    assert (env.depth_buffer.abs() <= 0.5 + 0.1 + 1e-6).all() and env.depth_buffer.std() > 0.05
    act = torch.zeros(4, 12)
    prev = env.depth_buffer[:, -1].clone()
    for _ in range(3):
        obs, _, rew, done, extras, *_ = env.step(act)
        assert extras["depth"].shape == (4, HC, WC)
        keep = done == 0
        assert torch.equal(extras["depth"][keep], prev[keep])                         # one step old
        assert not torch.equal(env.depth_buffer[:, -1], prev)
        prev = env.depth_buffer[:, -1].clone()
    env2 = cpu_env(4, seed=5)
    assert env2.step(act)[4]["depth"] is None                                         # use_camera off: no image


# ------------------------------------------------------------------ the student: modules, one DAgger update, learn_vision
def _vision_runner(env, tmp_path, device, steps=4):
    from quadrupedal_agility_amd.legged_gym.utils.helpers import class_to_dict
    from quadrupedal_agility_amd.tsc.legged_gym.envs.go2.go2_agility_config import Go2AgilityCfgPPO
    from quadrupedal_agility_amd.tsc.rsl_rl.runners import OnPolicyRunner
    tcfg = class_to_dict(Go2AgilityCfgPPO())
    tcfg["depth_encoder"]["if_depth"] = True
    tcfg["depth_encoder"]["num_steps_per_env"] = steps
    return OnPolicyRunner(env, tcfg, log_dir=str(tmp_path), device=device)


def test_depth_encoder_shapes_and_reference_parameter_names():
    """tsc/rsl_rl/modules/depth_backbone.py:7-109, byol.py: output = [latent 32 | headings 2 | softmax class 6]; the GRU state carries
    over calls; the state dict has the reference's keys (a reference depth_encoder_state_dict loads unchanged)"""
    from quadrupedal_agility_amd.tsc.legged_gym.envs.go2.go2_agility_config import Go2AgilityCfg
    from quadrupedal_agility_amd.tsc.rsl_rl.modules.depth_backbone import DepthOnlyFCBackbone58x87, RecurrentDepthBackbone
    torch.manual_seed(0)
    cfg = Go2AgilityCfg()
    enc = RecurrentDepthBackbone(DepthOnlyFCBackbone58x87(cfg.env.n_proprio, 32, 512), 32, cfg)
    img, prop = torch.rand(3, HC, WC) - 0.5, torch.randn(3, cfg.env.n_proprio)
    out1 = enc(img, prop); h1 = enc.hidden_states.clone()
    out2 = enc(img, prop)
    assert out1.shape == (3, 32 + 2 + 6) and torch.allclose(out1[:, 34:].sum(1), torch.ones(3), atol=1e-6)
    assert not torch.allclose(out1, out2) and enc.hidden_states.shape == (1, 3, 512) and not torch.equal(h1, enc.hidden_states)
    keys = set(enc.state_dict())
    for k in ("base_backbone.image_compression.0.weight", "base_backbone.image_compression.6.weight", "combination_mlp.0.weight", "rnn.weight_ih_l0",
              "output_mlp.0.bias", "byol_learner.online_encoder.projector.0.weight", "byol_learner.online_encoder.projector.1.running_mean",
              "byol_learner.online_predictor.3.weight", "byol_learner.target_encoder.net.image_compression.0.weight",
              "byol_learner.target_encoder.projector.3.bias", "byol_learner.net.image_compression.3.weight"):
        assert k in keys, k
    assert enc.state_dict()["base_backbone.image_compression.6.weight"].shape == (128, 64 * 25 * 39)
    loss = enc.byol_learner(torch.rand(8, HC, WC) - 0.5)
    assert 0.0 <= loss.item() <= 8.0                                                     # 2 x (2 - 2 cos)
    before = [p.clone() for p in enc.byol_learner.target_encoder.parameters()]
    with torch.no_grad():
        for p in enc.byol_learner.online_encoder.parameters():
            p.add_(1.0)
    enc.byol_learner.update_moving_average()
    for b, p, q in zip(before, enc.byol_learner.target_encoder.parameters(), enc.byol_learner.online_encoder.parameters()):
        assert torch.allclose(p, 0.99 * b + 0.01 * q, atol=1e-6)                         # EMA 0.99 (:74-84)


def test_gaussian_blur_is_a_normalised_separable_3x3():
    from quadrupedal_agility_amd.tsc.rsl_rl.modules.byol import GaussianBlur3
    torch.manual_seed(0)
    x = torch.zeros(2, 9, 9); x[:, 4, 4] = 1.0
    y = GaussianBlur3((1.0, 1.0))(x)
    k = np.exp(-0.5 * np.array([-1.0, 0.0, 1.0]) ** 2); k /= k.sum()
    assert np.allclose(y[0, 3:6, 3:6].numpy(), np.outer(k, k), atol=1e-6) and abs(y[0].sum().item() - 1.0) < 1e-6
    assert torch.allclose(GaussianBlur3()(torch.full((2, 5, 7), 0.3)), torch.full((2, 5, 7), 0.3), atol=1e-6)      # reflect padding keeps a constant


def test_learn_vision_runs_on_cpu(tmp_path):
    """learn_vision (:278-441) end to end on the oracle twins: the student's nets move, the teacher and the behaviour controller do
    not, the checkpoint carries the two depth keys and loads back; a teacher-only checkpoint seeds the student with the teacher's actor"""
    import os
    from tests.test_tsc_course_env import cpu_env
    torch.manual_seed(1)
    env = cpu_env(6, seed=2, env__episode_length_s=0.5, depth__use_camera=True)
    runner = _vision_runner(env, tmp_path, "cpu")
    assert runner.learn == runner.learn_vision
    snap = lambda m: {k: v.clone() for k, v in m.state_dict().items()}                  # noqa: E731
    enc0, act0, teacher0, bbc0 = snap(runner.alg.depth_encoder), snap(runner.alg.depth_actor), snap(runner.alg.actor_critic), snap(runner.actor_critic_bbc)
    runner.learn(2)
    moved = lambda a, m: any(not torch.equal(a[k], v) for k, v in m.state_dict().items())   # noqa: E731
    assert moved(enc0, runner.alg.depth_encoder) and moved(act0, runner.alg.depth_actor)
    assert not moved(teacher0, runner.alg.actor_critic) and not moved(bbc0, runner.actor_critic_bbc)
    assert all(np.isfinite(v) for v in runner.last_vision.values()) and env.bk._cfg.next_goal_threshold == pytest.approx(0.45)
    assert runner.alg.depth_actor_optimizer.param_groups[0]["lr"] == pytest.approx(1e-3 - (1e-3 - 1e-5) * 1 / 20000)
    ck = torch.load(os.path.join(str(tmp_path), "model.pt"), weights_only=False)
    assert {"depth_encoder_state_dict", "depth_actor_state_dict", "model_state_dict", "estimator_state_dict"} <= set(ck)
    runner.load(os.path.join(str(tmp_path), "model.pt"))
    del ck["depth_encoder_state_dict"], ck["depth_actor_state_dict"]
    torch.save(ck, os.path.join(str(tmp_path), "teacher.pt"))
    runner.load(os.path.join(str(tmp_path), "teacher.pt"))
    assert all(torch.equal(v, runner.alg.actor_critic.actor.state_dict()[k]) for k, v in runner.alg.depth_actor.state_dict().items())


@pytest.mark.gpu
def test_learn_vision_runs_on_gpu(tmp_path):
    """the reference's student configuration: 256 camera envs, 24 steps per update, depth images from the HIP ray-cast"""
    from quadrupedal_agility_amd.tsc.legged_gym.envs.base import legged_robot as lr
    from tests.test_tsc_course_env import make_cfg
    torch.manual_seed(0)
    cfg = make_cfg(256, 1, env__episode_length_s=2.0, depth__use_camera=True)
    env = lr.LeggedRobot(cfg, sim_device="cuda:0")
    runner = _vision_runner(env, tmp_path, "cuda:0", steps=24)
    losses = []
    for _ in range(4):
        runner.learn(1)
        losses.append(dict(runner.last_vision))
    print("learn_vision:", losses[0], "->", losses[-1], runner.last_perf)
    assert all(np.isfinite(v) for d in losses for v in d.values())
    assert losses[-1]["obst_type_loss"] < losses[0]["obst_type_loss"] + 0.05          # the class head starts learning at once
    assert env.depth_buffer.std() > 0.05 and runner.last_perf["fps"] > 1e3


@pytest.mark.gpu
@pytest.mark.parametrize("calls", [(3,), (2, 1, 1)])
def test_recorded_vision_env_step_equals_the_eager_one(tmp_path, monkeypatch, calls):
    """(`calls` = (2, 1, 1), ADVICE r5: learn() entered three times, as bench.py --tsc --vision and a resumed training do -- the replays of the
    later calls must read the action history and behaviour observation THAT call uses, i.e. the persistent ones.)

    r5: the env half of a `learn_vision` step (set_commands -> behaviour policy -> env.step -> next behaviour observation, action-history
    restart) is recorded once per camera phase and replayed (`OnPolicyRunner._vision_env_step`); the student's networks stay eager.  Same seeds,
    3 iterations x 24 steps, depth.update_interval 2 (both camera phases recorded and replayed many times): the robots' state, the depth images and the
    student's weights after the third update must equal the all-eager run's (env state and images bit for bit; the weights to 1e-6: the
    update is eager torch in both runs)."""
    import random
    from quadrupedal_agility_amd.tsc.legged_gym.envs.base import legged_robot as lr
    from tests.test_tsc_course_env import make_cfg
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("QA_TSC_ROLLOUT_GRAPH", mode)
        torch.manual_seed(0); random.seed(0); np.random.seed(0)
        cfg = make_cfg(128, 1, env__episode_length_s=1.0, depth__use_camera=True, depth__update_interval=2)
        env = lr.LeggedRobot(cfg, sim_device="cuda:0")
        runner = _vision_runner(env, tmp_path / mode, "cuda:0", steps=24)
        runner.log_dir = None                                   # no logging: the bench's mode
        for n in calls:
            runner.learn(n)
        torch.cuda.synchronize()
        if mode == "1":
            g = runner._vs["graphs"]
            assert set(g) == {False, True} and all(v is not False for v in g.values()), g          # both camera phases were recorded
        res[mode] = dict(root=env.root_states.clone(), depth=env.depth_buffer.clone(), obs=env.get_observations().clone(),
                         enc={k: v.clone() for k, v in runner.alg.depth_encoder.state_dict().items()},
                         act={k: v.clone() for k, v in runner.alg.depth_actor.state_dict().items()}, losses=dict(runner.last_vision))
    a, b = res["0"], res["1"]
    assert torch.equal(a["root"], b["root"]) and torch.equal(a["depth"], b["depth"]) and torch.equal(a["obs"], b["obs"])
    for part in ("enc", "act"):
        for k in a[part]:
            assert torch.allclose(a[part][k].float(), b[part][k].float(), rtol=1e-6, atol=1e-6), (part, k)
    assert all(abs(a["losses"][k] - b["losses"][k]) <= 1e-5 * max(1.0, abs(a["losses"][k])) for k in a["losses"])
