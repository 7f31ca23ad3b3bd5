"""Parity of the HIP path (through the C ABI) with the CPU oracle and with the recorded known
answers.  Bars: primary hit IDs bit-exact; colours within 1/255 per channel on >= 99.9 % of the
pixels (the packet path has no transcendental arithmetic, so it is held to bit-exact pixels);
t/u/v of the traversal operators bit-exact.
"""
import numpy as np
import pytest

import oracle_lib
from conftest import POSES, bits, colour_mismatch_fraction, pose_cam
from raytracing_amd import MODE_PACKET, MODE_RECURSIVE, Camera, MrtError, Raytracer, binding, scene_io

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rt512(maze_scene):
    rt = Raytracer(512, 512, device=0)
    rt.build_bvh(maze_scene)
    yield rt
    rt.close()


def _check_frame(rt, ora, cam, W, H, mode, spp_grid=1, exact_pixels=True):
    fb, ids = rt.render(cam, mode=mode, spp_grid=spp_grid)
    ref = ora.render(cam, W, H, mode=mode, spp_grid=spp_grid)
    assert np.array_equal(ids, ref["ids"]), "hit IDs differ on %d rays" % int((ids != ref["ids"]).sum())
    frac = colour_mismatch_fraction(fb, ref["fb"])
    assert frac <= 0.001, "colours beyond 1/255 on %.4f %% of pixels" % (100 * frac)
    if exact_pixels:
        assert np.array_equal(fb, ref["fb"]), "%d pixels differ" % int((fb != ref["fb"]).sum())
    return fb, ids, ref


@pytest.mark.parametrize("pose", POSES, ids=[p[0] for p in POSES])
def test_packet_frames_match_oracle(rt512, maze_oracle, pose):
    _check_frame(rt512, maze_oracle, pose_cam(pose), 512, 512, MODE_PACKET)


@pytest.mark.parametrize("pose", POSES, ids=[p[0] for p in POSES])
def test_recursive_frames_match_oracle(rt512, maze_oracle, pose):
    fb, ids, ref = _check_frame(rt512, maze_oracle, pose_cam(pose), 512, 512, MODE_RECURSIVE, exact_pixels=False)
    counts = rt512.ray_counts()
    c = maze_oracle.render(pose_cam(pose), 512, 512, mode=MODE_RECURSIVE, want_counters=True)["counters"]
    assert counts["primary"] == c["primary_rays"] == 512 * 512
    assert counts["shadow"] == c["shadow_rays"] and counts["secondary"] == c["secondary_rays"]
    assert counts["primary_hits"] == int((ids >= 0).sum())


@pytest.mark.parametrize("i", range(4))
def test_known_answer_frames_on_gpu(maze_scene, known, i):
    """The recorded whole-frame answers of the reference, reproduced by the GPU alone (no oracle)."""
    k = known["packet_frames"][i]
    rt = Raytracer(k["W"], k["H"], device=0)
    rt.build_bvh(maze_scene)
    cam = Camera()
    if k["pose"] is not None:
        cam.set_position(k["pose"]["pos"])
        cam.look_at(k["pose"]["target"])
    fb, ids = rt.render(cam, mode=MODE_PACKET)
    counts = rt.ray_counts()
    rt.close()
    assert int((ids >= 0).sum()) == k["hits"] == counts["primary_hits"] == counts["shadow"]
    assert oracle_lib.idsum(ids) == k["idsum"]
    assert "%016x" % oracle_lib.fbhash(fb) == k["fbhash"]


def test_known_answer_rays_on_gpu(maze_scene, known):
    rt = Raytracer(800, 600, device=0)
    rt.build_bvh(maze_scene)
    fb, ids = rt.render(oracle_lib.IDENTITY, mode=MODE_PACKET)
    rt.close()
    for k in known["rays_800x600_P0"]:
        x, y = k["px"]
        assert int(ids[y, x]) == k["hit"] and "%08x" % fb[y, x] == k["pixel"]


def test_full_hd_both_modes_match_oracle(maze_scene, maze_oracle):
    """The headline configuration's size: 1920x1080, packet and recursive."""
    rt = Raytracer(1920, 1080, device=0)
    rt.build_bvh(maze_scene)
    for pose in (POSES[1], POSES[0]):       # POSES[0] is the default camera, the benchmark's pose
        cam = pose_cam(pose)
        _check_frame(rt, maze_oracle, cam, 1920, 1080, MODE_PACKET)
        _check_frame(rt, maze_oracle, cam, 1920, 1080, MODE_RECURSIVE, exact_pixels=False)
    rt.close()


def _random_rays(rng, n, scene_lo, scene_hi):
    O = rng.uniform(scene_lo, scene_hi, (n, 3)).astype(np.float32)
    D = rng.normal(size=(n, 3)).astype(np.float32)
    D /= np.linalg.norm(D, axis=1, keepdims=True).astype(np.float32)
    # axis-aligned and zero-component directions: 1/D = +-inf and 0*inf = NaN paths
    D[::17, 0] = 0.0
    D[::29, 1] = -0.0
    D[::31] = np.float32([0, -1, 0])
    D[5::37] = np.float32([0, 0, -1])
    return O, D.astype(np.float32)


def test_trace_packets_operator_bit_exact(rt512, maze_oracle):
    rng = np.random.RandomState(11)
    n = 96 * 64
    O, D = _random_rays(rng, n, (-18, -1, -20), (18, 6, 20))
    # coherent packets too: a common origin per packet
    O[: 48 * 64] = np.repeat(O[:48], 64, axis=0)
    with np.errstate(divide="ignore"):
        rD = (np.float32(1.0) / D).astype(np.float32)
    t = np.full(n, 1e8, np.float32)
    t[::7] = np.float32(rng.uniform(0.5, 10, len(t[::7])))
    g = rt512.trace_packets(O.reshape(-1, 64, 3), D.reshape(-1, 64, 3), rD.reshape(-1, 64, 3), t)
    r = maze_oracle.trace_packets(O.reshape(-1, 64, 3), D.reshape(-1, 64, 3), rD.reshape(-1, 64, 3), t)
    assert np.array_equal(g[1], r[1])
    assert np.array_equal(bits(g[0]), bits(r[0]))
    hit = r[1] >= 0
    assert hit.sum() > 500
    assert np.array_equal(bits(g[2])[hit], bits(r[2])[hit]) and np.array_equal(bits(g[3])[hit], bits(r[3])[hit])


def test_trace_rays_operator_bit_exact(rt512, maze_oracle, maze_scene):
    rng = np.random.RandomState(12)
    n = 20000
    O, D = _random_rays(rng, n, (-18, -1, -20), (18, 6, 20))
    # rays that start exactly on a surface, like the reference's shadow rays (no epsilon)
    m = maze_scene["meshes"][0]
    tri = m["tri"][rng.randint(0, len(m["tri"]), 4000)]
    w = rng.dirichlet((1, 1, 1), 4000).astype(np.float32)
    O[:4000] = (m["pos"][tri[:, 0]] * w[:, 0:1] + m["pos"][tri[:, 1]] * w[:, 1:2] + m["pos"][tri[:, 2]] * w[:, 2:3]).astype(np.float32)
    t = np.full(n, np.inf, np.float32)
    t[::5] = np.float32(1e8)
    g = rt512.trace_rays(O, D, t)
    r = maze_oracle.trace_rays(O, D, t)
    assert np.array_equal(g[1], r[1])
    assert np.array_equal(bits(g[0]), bits(r[0]))
    hit = r[1] >= 0
    assert hit.sum() > 1000
    assert np.array_equal(bits(g[2])[hit], bits(r[2])[hit]) and np.array_equal(bits(g[3])[hit], bits(r[3])[hit])


@pytest.mark.parametrize("g", [2, 4])
def test_supersampling_grid_matches_oracle(maze_scene, maze_oracle, g):
    rt = Raytracer(256, 128, device=0)
    rt.build_bvh(maze_scene)
    _check_frame(rt, maze_oracle, pose_cam(POSES[1]), 256, 128, MODE_PACKET, spp_grid=g)
    rt.close()


def _soup_with_materials(n, seed):
    s = scene_io.soup_scene(n, seed=seed)
    m = s["meshes"][0]
    # split into three meshes with different materials so reflection AND refraction both recurse
    third = (n // 3) * 3
    parts = []
    for j, (refl, refr, rindex) in enumerate(((0.0, 0.0, 1.0), (0.4, 0.0, 1.0), (0.2, 0.3, 1.5))):
        lo, hi = j * third // 3, ((j + 1) * third // 3 if j < 2 else n)
        v0, v1 = lo * 3, hi * 3
        parts.append({"pos": m["pos"][v0:v1], "uv": m["uv"][v0:v1], "norm": m["norm"][v0:v1], "N": m["N"][lo:hi],
                      "tri": (m["tri"][lo:hi] - v0).astype(np.int32), "texture": 0,
                      "bounds": np.concatenate([m["pos"][v0:v1].min(0), m["pos"][v0:v1].max(0)]),
                      "refl": refl, "refr": refr, "rindex": rindex})
    s["meshes"] = parts
    return s


@pytest.mark.parametrize("n,lights", [(3000, 1), (40000, 3), (500, 0)])
def test_soup_scenes_both_modes(n, lights):
    scene = _soup_with_materials(n, seed=n)
    ora = oracle_lib.Oracle()
    ora.add_scene(scene)
    ora.build_bvh()
    ora.set_threads(8)
    rt = Raytracer(256, 256, device=0)
    rt.build_bvh(scene)
    L = np.float32([[0, 5, 0, 1, 1, 1, 1], [10, 3, -20, 0, 0.5, 0.25, 1.0], [-8, -4, -10, 0, 0.2, 0.9, 0.3]])[:lights]
    ora.set_lights(L)
    rt.set_lights(L)
    for pose in (POSES[0], ("soup", (5, 1, 4), (0, 0, -20))):
        cam = pose_cam(pose)
        _check_frame(rt, ora, cam, 256, 256, MODE_PACKET)
        _check_frame(rt, ora, cam, 256, 256, MODE_RECURSIVE, exact_pixels=False)
    rt.close()


def test_equal_distances_keep_the_reference_order_of_a_leaf():
    """Every triangle of the scene exists eight times over, the copies shuffled: whole leaves of triangles that a ray hits
    at exactly the same distance.  The reference keeps the copy that comes first in its leaf (strict `T < t`,
    raytracer.cpp:1336-1344); the device arrays hold long leaves in another order and decide by the triangles' places in
    the reference's order (kernels.hip: Tie).  Hit ids must equal the oracle's in both shaders, also from poses whose
    tiles are handed out in sixteenths to teams of lanes."""
    rng = np.random.RandomState(5)
    base = scene_io.soup_scene(60, seed=77)
    m = base["meshes"][0]
    tri = np.asarray(m["tri"], np.int32).reshape(-1, 3)
    pos, uv, norm, N = (np.asarray(m[k], np.float32) for k in ("pos", "uv", "norm", "N"))
    pos = (pos * np.float32([0.2, 1.0, 0.2]) + np.float32([0, 0, -8])).astype(np.float32)      # (a compact cloud in front of both cameras)
    copies = 8
    order = rng.permutation(len(tri) * copies)
    src = order % len(tri)                                   # which original triangle each new triangle copies
    new_pos = pos[tri[src]].reshape(-1, 3)                   # three vertices of its own per copy
    new_uv = uv[tri[src]].reshape(-1, 2)
    new_norm = norm[tri[src]].reshape(-1, 3)
    new_tri = np.arange(3 * len(src), dtype=np.int32).reshape(-1, 3)
    scene = {"textures": base["textures"],
             "meshes": [{"pos": new_pos, "uv": new_uv, "norm": new_norm, "N": N[src], "tri": new_tri, "texture": 0,
                         "bounds": np.concatenate([new_pos.min(0), new_pos.max(0)]).astype(np.float32), "refl": 0.2, "refr": 0.3, "rindex": 1.5}]}
    ora = oracle_lib.Oracle()
    ora.add_scene(scene)
    info = ora.build_bvh()
    mm, fc, idx = ora.bvh_arrays()
    assert fc[:, 1].max() >= 16                              # equal centroids cannot be told apart: long leaves
    W, H = 256, 128
    rt = Raytracer(W, H, device=0)
    rt.build_bvh(scene)
    seen_ties = 0
    for cam in (pose_cam(POSES[0]), oracle_lib.lookat((1.5, 0.5, -2), (0, 0, -8))):
        for mode in (MODE_PACKET, MODE_RECURSIVE):
            for _ in range(3):                               # (the later frames of a view split their expensive tiles)
                fb, ids = rt.render(cam, mode=mode)
            want = ora.render(cam, W, H, mode=mode)
            assert np.array_equal(ids, want["ids"]), "%d hit ids differ" % int((ids != want["ids"]).sum())
            assert colour_mismatch_fraction(fb, want["fb"]) <= 0.001
            seen_ties += int((ids >= 0).sum())               # (every hit of this scene is a tie of eight)
    assert seen_ties > 2000
    rt.close()


@pytest.mark.parametrize("scale", [1e5, 3e-4, 1e9, 1e-9])
def test_scaled_scenes_exact_division_paths(scale):
    """The same soup blown up or shrunk, camera and lights with it.  Scales 1e5 and 3e-4 keep every
    coordinate inside the ranges where the single-ray traversal replaces divisions by corrected
    multiplications (and shadow rays decide boxes from interval bounds); 1e9 and 1e-9 leave them, so the
    true-division code runs.  Results must equal the oracle's either way."""
    scene = _soup_with_materials(4000, seed=11)
    s = np.float32(scale)
    for m in scene["meshes"]:
        m["pos"] = (np.asarray(m["pos"], np.float32) * s).astype(np.float32)
        if "bounds" in m:
            m["bounds"] = (np.asarray(m["bounds"], np.float32) * s).astype(np.float32)
    ora = oracle_lib.Oracle()
    ora.add_scene(scene)
    ora.build_bvh()
    ora.set_threads(8)
    rt = Raytracer(192, 128, device=0)
    rt.build_bvh(scene)
    L = np.float32([[0, 5, 0, 1, 1, 1, 1], [10, 3, -20, 0, 0.5, 0.25, 1.0]])
    L[:, :3] *= s
    ora.set_lights(L)
    rt.set_lights(L)
    for pos, target in (((0, 0, 0), (0, 0, -1)), ((5, 1, 4), (0, 0, -20))):
        cam = oracle_lib.lookat(tuple(np.float32(pos) * s), tuple(np.float32(target) * s))
        _check_frame(rt, ora, cam, 192, 128, MODE_PACKET)
        _check_frame(rt, ora, cam, 192, 128, MODE_RECURSIVE, exact_pixels=False)
    rt.close()


def test_tiny_and_empty_scenes():
    rt = Raytracer(64, 64, device=0)
    for n in (0, 1, 2, 3):
        scene = scene_io.soup_scene(max(n, 1), seed=3)
        if n == 0:
            scene["meshes"] = []
        ora = oracle_lib.Oracle()
        ora.add_scene(scene)
        ora.build_bvh()
        rt.build_bvh(scene)
        for mode in (MODE_PACKET, MODE_RECURSIVE):
            fb, ids, ref = _check_frame(rt, ora, oracle_lib.IDENTITY, 64, 64, mode)
            if n == 0:
                assert (fb == 0xFF4C4C4C).all() and (ids == -1).all()
    rt.close()


def test_row_bands_on_device_equal_full_frame(maze_scene):
    """mrt_render_device on disjoint tile-row bands (what each GPU of a multi-GPU run does)
    writes exactly the full frame's pixels."""
    import torch
    W, H = 512, 256
    rt = Raytracer(W, H, device=0)
    rt.build_bvh(maze_scene)
    cam = pose_cam(POSES[2])
    full, _ = rt.render(cam, mode=MODE_RECURSIVE)
    fb = torch.zeros((H, W), dtype=torch.int32, device="cuda:0")
    for y0, y1 in ((0, 64), (64, 72), (72, 256)):
        rt.render_device(cam, fb.data_ptr(), W, mode=MODE_RECURSIVE, rows=(y0, y1), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(fb.cpu().numpy().view(np.uint32), full)
    rt.close()


@pytest.mark.parametrize("mode,grid", [(MODE_RECURSIVE, 1), (MODE_PACKET, 2)])
def test_interleaved_shares_on_device_equal_full_frame(maze_scene, mode, grid):
    """mrt_render_device_part: the shares of an interleaved tile-row split, in place and packed,
    reassemble to exactly the full frame (hit ids included), also with a sample grid and with a
    ragged last cycle (33 tile rows)."""
    import torch
    from raytracing_amd.multigpu import part_row_index
    W, H = 256, 264
    rt = Raytracer(W, H, device=0)
    rt.build_bvh(maze_scene)
    cam = pose_cam(POSES[1])
    full, full_ids = rt.render(cam, mode=mode, spp_grid=grid)
    stream = torch.cuda.current_stream().cuda_stream
    for world, block in ((2, 1), (3, 4), (8, 2), (4, 40)):
        fb = torch.zeros((H, W), dtype=torch.int32, device="cuda:0")
        ids = torch.full((H * grid, W * grid), -7, dtype=torch.int32, device="cuda:0")
        for r in range(world):
            rt.render_device_part(cam, fb.data_ptr(), W, r, world, block, compact=False, mode=mode, spp_grid=grid,
                                  ids_ptr=ids.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        assert np.array_equal(fb.cpu().numpy().view(np.uint32), full)
        assert np.array_equal(ids.cpu().numpy(), full_ids)
        got = np.zeros((H, W), np.uint32)
        for r in range(world):
            rows = part_row_index(H, world, r, block)
            assert rt.L.mrt_part_rows(H, r, world, block) == len(rows)
            packed = torch.zeros((max(len(rows), 8), W), dtype=torch.int32, device="cuda:0")
            pids = torch.full((max(len(rows), 8) * grid, W * grid), -7, dtype=torch.int32, device="cuda:0")
            rt.render_device_part(cam, packed.data_ptr(), W, r, world, block, compact=True, mode=mode, spp_grid=grid,
                                  ids_ptr=pids.data_ptr(), stream=stream)
            torch.cuda.synchronize()
            got[rows] = packed.cpu().numpy().view(np.uint32)[:len(rows)]
            if len(rows):
                id_rows = (rows[:, None] * grid + np.arange(grid)[None, :]).reshape(-1)
                assert np.array_equal(pids.cpu().numpy()[:len(rows) * grid], full_ids[id_rows])
        assert np.array_equal(got, full)
    with pytest.raises(MrtError, match="share"):
        rt.render_device_part(cam, fb.data_ptr(), W, 2, 2, 1)
    rt.close()


@pytest.mark.parametrize("world,block,pitch_extra", [(2, 1, 0), (3, 4, 0), (8, 2, 8), (4, 40, 3)])
def test_native_gather_scatter_reassembles_the_frame(maze_scene, world, block, pitch_extra):
    """The native exchange (libmrt_gather.so) without its transport: every share but rank 0's is rendered packed,
    copied into rank 0's receive buffer for that peer and put in place by the scatter kernel; the frame must equal
    the undivided one.  Only ncclSend / ncclRecv themselves are not exercised (one GPU here)."""
    import torch
    from raytracing_amd.multigpu import NativeGather
    W, H = 256, 264
    rt = Raytracer(W, H, device=0)
    rt.build_bvh(maze_scene)
    cam = pose_cam(POSES[1])
    full, _ = rt.render(cam, mode=MODE_RECURSIVE)
    stream = torch.cuda.current_stream().cuda_stream
    pitch = W + pitch_extra
    frame = torch.zeros((H, pitch), dtype=torch.int32, device="cuda:0")
    g = NativeGather(W, H, 0, world, block)
    rt.render_device_part(cam, frame.data_ptr(), pitch, 0, world, block, compact=False, mode=MODE_RECURSIVE, stream=stream)
    for r in range(1, world):
        n = g.share_rows(r)
        assert n == rt.L.mrt_part_rows(H, r, world, block)
        if n == 0:
            assert not g.staging(r)
            continue
        packed = torch.zeros((n, W), dtype=torch.int32, device="cuda:0")
        rt.render_device_part(cam, packed.data_ptr(), W, r, world, block, compact=True, mode=MODE_RECURSIVE, stream=stream)
        g.stage_local(r, packed.data_ptr(), stream)
        g.scatter(r, frame.data_ptr(), pitch, stream)
    torch.cuda.synchronize()
    assert np.array_equal(frame[:, :W].cpu().numpy().view(np.uint32), full)
    if pitch_extra:
        assert int(frame[:, W:].abs().sum()) == 0
    g.close()
    rt.close()


def test_native_exchange_with_its_own_communicator_one_rank(maze_scene):
    """multigpu.NativeExchange, the transport bench.py --gpus N drives: communicator id, communicator (RCCL, one rank:
    all a one-GPU box can hold), whole-frame gather from a pitched frame and the share exchange of a frame of one share,
    both queued on a side stream.  The many-rank forms of the same calls are rehearsed without RCCL by the tests above."""
    import torch
    from raytracing_amd.multigpu import NativeExchange

    class OneRank:
        def broadcast_object_list(self, box, src=0):
            pass

    W, H = 256, 128
    rt = Raytracer(W, H, device=0)
    rt.build_bvh(maze_scene)
    cam = pose_cam(POSES[1])
    want, _ = rt.render(cam, mode=MODE_PACKET)
    nx = NativeExchange(W, H, 0, 1, OneRank())
    side = torch.cuda.Stream()
    fb = torch.zeros((H, W + 64), dtype=torch.int32, device="cuda:0")
    frames = torch.full((1, H, W), -1, dtype=torch.int32, device="cuda:0")
    main = torch.cuda.current_stream()
    rt.render_device(cam, fb.data_ptr(), W + 64, mode=MODE_PACKET, stream=main.cuda_stream)
    rendered = torch.cuda.Event()
    rendered.record(main)
    side.wait_event(rendered)
    nx.frames(fb.data_ptr(), W + 64, frames.data_ptr(), side.cuda_stream)
    nx.rows(fb.data_ptr(), W + 64, side.cuda_stream)
    side.synchronize()
    assert np.array_equal(frames[0].cpu().numpy().view(np.uint32), want)
    assert np.array_equal(fb[:, :W].cpu().numpy().view(np.uint32), want)
    nx.close()
    rt.close()


def test_8k_16spp_frame_rows_match_oracle(maze_scene, maze_oracle, full_size):
    """The multi-GPU configuration at its full size (7680x4320, 4x4 sample grid = 530,841,600 primary
    rays): the whole frame is rendered once and its hash compared with the hash the oracle produced for this
    frame offline (tests/golden/full_size.json); 34 of its 540 8-row strips -- first, last, and 32 drawn at
    random -- are also compared with the oracle here, pixels and, for six of them, the hit ids of all 16 samples."""
    import torch
    W, H, g = 7680, 4320, 4
    rt = Raytracer(W, H, device=0)
    rt.build_bvh(maze_scene)
    cam = oracle_lib.IDENTITY
    stream = torch.cuda.current_stream().cuda_stream
    fb = torch.zeros((H, W), dtype=torch.int32, device="cuda:0")
    rt.render_device(cam, fb.data_ptr(), W, mode=MODE_PACKET, spp_grid=g, stream=stream)
    torch.cuda.synchronize()
    counts = rt.ray_counts()
    assert counts["primary"] == W * H * g * g and counts["shadow"] == counts["primary_hits"]
    gold = full_size["c4_maze_8k_16spp_packet"]
    assert counts["primary_hits"] == gold["primary_hits"] == gold["counters"]["shadow_rays"]
    whole = fb.cpu().numpy().view(np.uint32)
    assert "%016x" % oracle_lib.fbhash(whole) == gold["fbhash"], "the 8K / 16 spp frame differs from the oracle's"
    n_rows = H // 8
    strip = torch.zeros((8, W), dtype=torch.int32, device="cuda:0")
    ids = torch.zeros((8 * g, W * g), dtype=torch.int32, device="cuda:0")
    maze_oracle.set_threads(16)
    rows = [0, n_rows - 1] + sorted(np.random.RandomState(5).choice(np.arange(1, n_rows - 1), 32, replace=False).tolist())
    for j, R in enumerate(rows):
        ref_fb, ref_ids = maze_oracle.render_band(cam, W, H, 8 * R, 8 * R + 8, mode=MODE_PACKET, spp_grid=g)
        assert np.array_equal(whole[8 * R:8 * R + 8], ref_fb), "tile row %d" % R
        if j % 6 == 0:
            # the same strip as a packed share (one tile row per share), with the hit ids of its samples
            rt.render_device_part(cam, strip.data_ptr(), W, R, n_rows, 1, compact=True, mode=MODE_PACKET, spp_grid=g,
                                  ids_ptr=ids.data_ptr(), stream=stream)
            torch.cuda.synchronize()
            assert np.array_equal(strip.cpu().numpy().view(np.uint32), ref_fb)
            assert np.array_equal(ids.cpu().numpy(), ref_ids)
    rt.close()


def test_million_triangle_scene_4k_rows_match_oracle(maze_scene, full_size):
    """The large-scene and the rasterizer configurations at their full size: 1,000,000-triangle soup with the grass
    texture, 3840x2160, scene in global memory.  BVH arrays identical to the oracle's; the whole packet frame (hash and
    hit-id checksum) and the whole recursive frame equal what the oracle produced offline (tests/golden/full_size.json);
    34 strips of each are also compared with the oracle here (hit ids and pixels)."""
    import torch
    W, H = 3840, 2160
    scene = scene_io.large_scene(maze_scene)
    ora = oracle_lib.Oracle()
    ora.add_scene(scene)
    ora.build_bvh()
    mm, fc, idx = ora.bvh_arrays()
    mine = binding.bvh_build_host(scene)
    n = ora.bvh_info["poolPtr"] + 1
    assert mine["poolPtr"] == ora.bvh_info["poolPtr"] and mine["maxDepth"] == ora.bvh_info["maxDepth"]
    assert np.array_equal(idx, mine["tri_idx"])
    assert np.array_equal(fc[3:], mine["node_first_cnt"][3:n])          # slots 1 and 2 are never written
    assert np.array_equal(bits(mm)[3:], bits(mine["node_minmax"][3:n]))
    rt = Raytracer(W, H, device=0)
    rt.build_bvh(scene)
    cam = oracle_lib.IDENTITY
    stream = torch.cuda.current_stream().cuda_stream
    n_rows = H // 8
    strip = torch.zeros((8, W), dtype=torch.int32, device="cuda:0")
    ids = torch.zeros((8, W), dtype=torch.int32, device="cuda:0")
    ora.set_threads(16)
    whole = torch.zeros((H, W), dtype=torch.int32, device="cuda:0")
    whole_ids = torch.zeros((H, W), dtype=torch.int32, device="cuda:0")
    for mode, key in ((MODE_PACKET, "c3_soup1m_4k_packet"), (MODE_RECURSIVE, "c3_soup1m_4k_recursive")):
        rt.render_device(cam, whole.data_ptr(), W, mode=mode, ids_ptr=whole_ids.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        gold = full_size[key]
        assert oracle_lib.idsum(whole_ids.cpu().numpy()) == gold["idsum"] and int((whole_ids >= 0).sum()) == gold["primary_hits"]
        if mode == MODE_PACKET:
            assert "%016x" % oracle_lib.fbhash(whole.cpu().numpy().view(np.uint32)) == gold["fbhash"]
        rows = [0, n_rows - 1] + sorted(np.random.RandomState(7 + mode).choice(np.arange(1, n_rows - 1), 32, replace=False).tolist())
        for R in rows:
            ref_fb, ref_ids = ora.render_band(cam, W, H, 8 * R, 8 * R + 8, mode=mode)
            rt.render_device_part(cam, strip.data_ptr(), W, R, n_rows, 1, compact=True, mode=mode,
                                  ids_ptr=ids.data_ptr(), stream=stream)
            torch.cuda.synchronize()
            assert np.array_equal(ids.cpu().numpy(), ref_ids), "mode %d tile row %d" % (mode, R)
            assert colour_mismatch_fraction(strip.cpu().numpy().view(np.uint32), ref_fb) <= 0.001
    # the rasterizer configuration on the same scene and size: coverage, depth winner, colours, 1/z
    from test_gpu_raster import _check
    covered = _check(rt, ora, cam, W, H)
    assert covered > 0.99
    rt.close()


def test_page_locked_caller_framebuffer(maze_scene):
    """mrt_render into memory from mrt_host_alloc (written by the GPU directly, also with a pitch wider
    than the frame) gives the same frame as into an ordinary buffer (staged)."""
    from raytracing_amd.binding import HostFrame
    W, H = 256, 128
    rt = Raytracer(W, H, device=0)
    rt.build_bvh(maze_scene)
    cam = pose_cam(POSES[1])
    ref, _ = rt.render(cam, mode=MODE_RECURSIVE)
    hf = HostFrame(H, W)
    rt.render(cam, mode=MODE_RECURSIVE, fb=hf.array, want_ids=False)
    assert np.array_equal(hf.array, ref)
    wide = HostFrame(H, W + 64)
    wide.array[:] = 0xDEADBEEF
    rt.render(cam, mode=MODE_RECURSIVE, fb=wide.array[:, :W], want_ids=False)
    assert np.array_equal(wide.array[:, :W], ref) and (wide.array[:, W:] == 0xDEADBEEF).all()
    hf.close(); wide.close()
    rt.close()


def test_root_gate_quirk_on_gpu(maze_scene, known):
    """The recorded effect of the reference's root-gate quirk, reproduced by the GPU alone: number of
    800x600 primary rays whose hit differs between the packet path (no gate) and the recursive shader."""
    rt = Raytracer(800, 600, device=0)
    rt.build_bvh(maze_scene)
    for k in known["root_gate_800x600"]:
        cam = Camera()
        if k["pose"] is not None:
            cam.set_position(k["pose"]["pos"])
            cam.look_at(k["pose"]["target"])
        _, a = rt.render(cam, mode=MODE_PACKET)
        _, b = rt.render(cam, mode=MODE_RECURSIVE)
        assert int((a != b).sum()) == k["rays_changed_by_gate"] == int(((a >= 0) & (b < 0)).sum())
    rt.close()


def test_context_lifecycle_releases_device_memory(maze_scene):
    """Creating, using (both ray modes, rasterizer, operators) and destroying contexts over and over leaves
    device memory where it was."""
    import torch
    torch.cuda.synchronize()

    def cycle():
        rt = Raytracer(256, 128, device=0)
        rt.build_bvh(maze_scene)
        rt.render(oracle_lib.IDENTITY, mode=MODE_PACKET)
        rt.render(oracle_lib.IDENTITY, mode=MODE_RECURSIVE, spp_grid=2)
        rt.raster(oracle_lib.IDENTITY)
        O = np.zeros((64, 3), np.float32)
        D = np.tile(np.float32([0, 0, -1]), (64, 1))
        rt.trace_rays(O, D, np.full(64, 1e8, np.float32))
        rt.close()

    for _ in range(3):
        cycle()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(20):
        cycle()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < 8 << 20, "device memory shrank by %d bytes over 20 context lifetimes" % (free0 - free1)


def test_launches_on_two_streams_do_not_disturb_each_other(maze_scene):
    """Frames of one context launched back to back on different streams: the context's tile queues and
    counters are shared, so the library orders the launches itself; both frames come out right."""
    import torch
    W, H = 512, 256
    rt = Raytracer(W, H, device=0)
    rt.build_bvh(maze_scene)
    cams = [pose_cam(POSES[1]), pose_cam(POSES[2])]
    refs = [rt.render(c, mode=MODE_RECURSIVE)[0] for c in cams]
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    fa = torch.zeros((H, W), dtype=torch.int32, device="cuda:0")
    fb = torch.zeros((H, W), dtype=torch.int32, device="cuda:0")
    torch.cuda.synchronize()
    for _ in range(5):
        rt.render_device(cams[0], fa.data_ptr(), W, mode=MODE_RECURSIVE, stream=sa.cuda_stream)
        rt.render_device(cams[1], fb.data_ptr(), W, mode=MODE_RECURSIVE, stream=sb.cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(fa.cpu().numpy().view(np.uint32), refs[0])
    assert np.array_equal(fb.cpu().numpy().view(np.uint32), refs[1])
    rt.close()


def test_packet_frames_follow_the_eye_and_the_scene(maze_scene, maze_oracle):
    """The packet path reads nodes and triangles relative to the eye from arrays the library rewrites only when the
    eye or the scene has changed.  One context, frames on alternating streams: the same eye with another view
    direction, another eye, the first eye again, then other scenes under an eye that stays — every frame equals the
    oracle's."""
    import torch
    W, H = 256, 128
    rt = Raytracer(W, H, device=0)
    rt.build_bvh(maze_scene)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    fb = torch.zeros((H, W), dtype=torch.int32, device="cuda:0")
    eye_a, eye_b = (0.0, 0.0, 0.0), (2.5, 0.5, -3.0)
    views = [("a", eye_a, (0, 0, -1)), ("a turned", eye_a, (1, 0.2, -0.3)), ("b", eye_b, (-1, 0, -1)), ("a again", eye_a, (0, 0, -1)),
             ("b turned", eye_b, (0.3, -0.2, 1))]
    for k, pose in enumerate(views):
        cam = np.asarray(pose_cam(pose), np.float32)
        rt.render_device(cam, fb.data_ptr(), W, mode=MODE_PACKET, stream=streams[k % 2].cuda_stream)
        torch.cuda.synchronize()
        want = maze_oracle.render(cam, W, H, mode=MODE_PACKET)
        assert np.array_equal(fb.cpu().numpy().view(np.uint32), want["fb"]), pose[0]
    cam = np.asarray(pose_cam(("soup", (5, 1, 4), (0, 0, -20))), np.float32)
    for seed in (3, 4):                         # other scenes of the same size, same eye
        soup = scene_io.soup_scene(2000, seed=seed)
        rt.build_bvh(soup)
        ora = oracle_lib.Oracle()
        ora.add_scene(soup)
        ora.build_bvh()
        got, ids = rt.render(cam, mode=MODE_PACKET)
        want = ora.render(cam, W, H, mode=MODE_PACKET)
        assert np.array_equal(ids, want["ids"]) and np.array_equal(got, want["fb"])
    rt.close()


def test_rays_in_the_plane_of_a_wall(maze_scene):
    """Rays that run inside a wall's plane are where a triangle test's verdict is rounding noise, and where skipping
    groups of a long leaf by their boxes must stand back (kernels.hip: not_grazing): lights placed exactly in the
    planes of the three largest axis-parallel walls (shadow rays graze them), and eyes placed in those planes looking
    along them.  Hit ids and packet pixels equal the oracle's; recursive pixels to 1/255."""
    f = np.float32
    tris = []
    for m in maze_scene["meshes"]:
        pos = np.asarray(m["pos"], f).reshape(-1, 3)
        tris.append(pos[np.asarray(m["tri"], np.int32).reshape(-1, 3)])
    T = np.concatenate(tris).astype(np.float64)
    n = np.cross(T[:, 1] - T[:, 0], T[:, 2] - T[:, 0])
    area = np.linalg.norm(n, axis=1)
    walls = []
    for axis in range(3):
        flat = (np.abs(n[:, axis]) == area) & (area > 0)          # normal exactly along this axis
        k = int(np.argmax(np.where(flat, area, 0)))
        assert flat[k]
        walls.append((axis, T[k]))
    W, H = 256, 128
    rt = Raytracer(W, H, device=0)
    rt.build_bvh(maze_scene)
    ora = oracle_lib.Oracle()
    ora.add_scene(maze_scene)
    ora.build_bvh()
    ora.set_threads(8)
    for axis, tri in walls:
        centre = tri.mean(0)
        along = tri[1] - tri[0]
        along /= np.linalg.norm(along)
        in_plane = centre + 2.5 * along                             # a point of the wall's plane, beside the triangle
        lights = np.array([[*in_plane, 1, 1, 1, 1]], f)
        rt.set_lights(lights)
        ora.set_lights(lights)
        eye = centre - 3.0 * along
        up_axis = (axis + 1) % 3
        eye[up_axis] += 0.25
        for cam in (pose_cam(POSES[0]), oracle_lib.lookat(tuple(eye), tuple(centre + 4.0 * along))):
            cam = np.asarray(cam, f)
            for mode in (MODE_PACKET, MODE_RECURSIVE):
                fb, ids = rt.render(cam, mode=mode)
                want = ora.render(cam, W, H, mode=mode)
                assert np.array_equal(ids, want["ids"]), "hit ids differ on %d rays (axis %d)" % (int((ids != want["ids"]).sum()), axis)
                if mode == MODE_PACKET:
                    assert np.array_equal(fb, want["fb"])
                else:
                    assert colour_mismatch_fraction(fb, want["fb"]) <= 0.001
    rt.close()


@pytest.mark.parametrize("W,H", [(1000, 600), (16, 8), (256, 8)])
def test_tile_order_from_previous_frame_changes_nothing(maze_scene, W, H):
    """The persistent kernel starts the tiles that were expensive in the previous launch of the same
    shape first.  Pixels, hit IDs and ray counters must not depend on that order: frames of a
    context that has a history (same pose again, then another pose) equal the frames rendered with
    the ordering switched off (MRT_FLAG_FRAME_ORDER)."""
    rt = Raytracer(W, H, device=0)
    rt.build_bvh(maze_scene)
    cams = [pose_cam(POSES[0]), pose_cam(POSES[2])]
    base = []
    for cam in cams:
        fb, ids = rt.render(cam, mode=MODE_RECURSIVE, extra_flags=binding.FLAG_FRAME_ORDER)
        base.append((fb.copy(), ids.copy(), rt.ray_counts()))
    for which in (0, 0, 0, 1, 1, 0):
        fb, ids = rt.render(cams[which], mode=MODE_RECURSIVE)
        assert np.array_equal(ids, base[which][1]) and np.array_equal(fb, base[which][0])
        assert rt.ray_counts() == base[which][2]
    rt.close()


@pytest.mark.parametrize("W,H", [(1920, 1080), (800, 600), (64, 40)])
def test_tile_order_of_a_walking_camera_changes_nothing(maze_scene, W, H):
    """While the camera moves from frame to frame the order of a frame's tiles also looks at the previous costs of the
    tiles around each tile (kernels.hip: tile_spread_kernel), and small frames split tiles whose costs come from another
    view.  Twelve frames of a camera walking sideways, then turning, equal the frames rendered with the ordering
    switched off: pixels, hit IDs, ray counters."""
    from raytracing_amd import Camera
    rt = Raytracer(W, H, device=0)
    rt.build_bvh(maze_scene)
    plain = Raytracer(W, H, device=0)
    plain.build_bvh(maze_scene)
    cam = Camera()
    for k in range(12):
        if k < 8:
            x, y, z = cam.get_position()
            cam.set_position((x + 0.16, y, z))
        else:
            cam.look_at((np.float32(0.3 * (k - 7)), np.float32(0.1), np.float32(-1.0)))
        fb, ids = rt.render(cam, mode=MODE_RECURSIVE)
        want_fb, want_ids = plain.render(cam, mode=MODE_RECURSIVE, extra_flags=binding.FLAG_FRAME_ORDER)
        assert np.array_equal(ids, want_ids) and np.array_equal(fb, want_fb), "frame %d" % k
        assert rt.ray_counts() == plain.ray_counts()
    rt.close()
    plain.close()


@pytest.mark.parametrize("W,H", [(512, 512), (800, 600), (256, 144)])
def test_expensive_tiles_in_sixteenths_change_nothing(maze_scene, maze_oracle, W, H):
    """Small frames of the recursive shader: from the second frame of a view on, the most expensive tiles of the previous
    frame are handed out as sixteen 2x2-pixel parts, every pixel of a part traced by a team of 16 lanes that share out the
    triangles of each leaf (kernels.hip: render_tile, traverse_nearest).  mrt_get_tile_schedule shows that it happens;
    pixels, hit ids and ray counts must be those of the frame rendered in frame order, and of the oracle."""
    rt = Raytracer(W, H, device=0)
    rt.build_bvh(maze_scene)
    n_tiles = (W // 8) * (H // 8)
    for pose in (POSES[0], POSES[1], POSES[4]):
        cam = pose_cam(pose)
        base_fb, base_ids = rt.render(cam, mode=MODE_RECURSIVE, extra_flags=binding.FLAG_FRAME_ORDER)
        assert rt.tile_schedule() == (n_tiles, n_tiles)
        base_counts = rt.ray_counts()
        parts = []
        for _ in range(5):
            fb, ids = rt.render(cam, mode=MODE_RECURSIVE)
            tiles, handed = rt.tile_schedule()
            assert tiles == n_tiles and handed >= tiles and (handed - tiles) % 15 == 0
            parts.append((handed - tiles) // 15)
            assert np.array_equal(ids, base_ids) and np.array_equal(fb, base_fb), "%s, frame with %d tiles in parts" % (pose[0], parts[-1])
            assert rt.ray_counts() == base_counts
        if pose[0] == "P0_identity":
            assert parts[0] == 0 and max(parts[1:]) > 0, parts        # (the first frame of a view has no costs to go by)
        want = maze_oracle.render(cam, W, H, mode=MODE_RECURSIVE)
        assert np.array_equal(base_ids, want["ids"]) and colour_mismatch_fraction(base_fb, want["fb"]) <= 0.001
    # the packet path and frames with a sample grid are never split
    rt.render(pose_cam(POSES[0]), mode=MODE_PACKET)
    rt.render(pose_cam(POSES[0]), mode=MODE_PACKET)
    assert rt.tile_schedule() == (n_tiles, n_tiles)
    rt.close()


def test_submit_and_wait_keep_frames_in_flight(maze_scene):
    """mrt_render_submit / mrt_render_wait: four frames queued at once land in their page-locked frames
    identical to what the synchronous mrt_render gives; a fifth is refused until one has been waited for."""
    from raytracing_amd.binding import HostFrame
    W, H = 640, 360
    rt = Raytracer(W, H, device=0)
    rt.build_bvh(maze_scene)
    cams = [pose_cam(p) for p in POSES[:4]]
    want = [rt.render(c, mode=MODE_RECURSIVE, want_ids=False)[0].copy() for c in cams]
    frames = [HostFrame(H, W) for _ in cams]
    tickets = [rt.submit(c, f.array, mode=MODE_RECURSIVE) for c, f in zip(cams, frames)]
    assert tickets == sorted(tickets) and len(set(tickets)) == 4
    with pytest.raises(MrtError):
        rt.submit(cams[0], frames[0].array, mode=MODE_RECURSIVE)          # MRT_MAX_IN_FLIGHT reached
    for t, f, w in zip(tickets, frames, want):
        rt.wait(t)
        assert np.array_equal(f.array, w)
    with pytest.raises(MrtError):
        rt.wait(tickets[0])                                               # waited for already
    with pytest.raises(MrtError):
        rt.submit(cams[0], np.zeros((H, W), np.uint32), mode=MODE_PACKET)  # pageable memory
    t = rt.submit(cams[1], frames[0].array, mode=MODE_PACKET)             # slots are reusable
    rt.wait(t)
    assert np.array_equal(frames[0].array, rt.render(cams[1], mode=MODE_PACKET, want_ids=False)[0])
    for f in frames:
        f.close()
    rt.close()


def test_error_paths(maze_scene):
    rt = Raytracer(64, 64, device=0)
    with pytest.raises(MrtError, match="scene_upload"):
        rt.render(oracle_lib.IDENTITY)
    rt.build_bvh(maze_scene)
    with pytest.raises(MrtError, match="grid"):
        rt.render(oracle_lib.IDENTITY, spp_grid=3)
    with pytest.raises(MrtError, match="mode"):
        rt.render(oracle_lib.IDENTITY, mode=7)
    for tex in (9, -1):        # outside the texture array; none at all (the tracer needs one on every mesh)
        bad = {"textures": maze_scene["textures"], "meshes": [dict(maze_scene["meshes"][0], texture=tex)]}
        with pytest.raises(MrtError, match="texture"):
            rt.build_bvh(bad)
    rt.build_bvh(maze_scene)
    with pytest.raises(MrtError, match="flag"):
        rt.render(oracle_lib.IDENTITY, extra_flags=1 << 20)
    # a pitch smaller than the width is refused before anything is launched or written, for pageable frames too
    fb = np.zeros((64, 64), np.uint32)
    rc = rt.L.mrt_render(rt.h, binding._p(oracle_lib.IDENTITY), 0, binding._p(fb), 32, None)
    assert rc == -1 and b"pitch" in rt.L.mrt_last_error(rt.h) and not fb.any()
    assert rt.L.mrt_render(rt.h, binding._p(oracle_lib.IDENTITY), 0, binding._p(fb), -64, None) == -1
    rt.close()


@pytest.mark.parametrize("loop", [["--frames", "2"], ["--frames", "5", "--pipeline"]], ids=["synchronous", "pipelined"])
def test_cpp_host_driver_renders_reference_frame(tmp_path, known, loop):
    """The headless C++ driver (Game::Init + Tick order through the host mirror classes) writes the
    frame whose hash the reference recorded: 800x600, default camera, packet path; also with the next
    frame kept in flight while the previous one is presented."""
    import os
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "raytracing_amd", "host", "mrt_game")
    out = str(tmp_path / "frame.ppm")
    subprocess.check_call([exe, "--scene", os.path.join(ROOT, "tests", "golden", "maze.mrts"), "--size", "800", "600",
                           "--mode", "packet", "--out", out] + loop)
    with open(out, "rb") as f:
        assert f.readline() == b"P6\n" and f.readline() == b"800 600\n" and f.readline() == b"255\n"
        rgb = np.frombuffer(f.read(), np.uint8).reshape(600, 800, 3).astype(np.uint32)
    rt = Raytracer(800, 600, device=0)
    rt.build_bvh(scene_io.load_scene(os.path.join(ROOT, "tests", "golden", "maze.mrts")))
    fb, _ = rt.render(oracle_lib.IDENTITY, mode=MODE_PACKET)
    rt.close()
    assert "%016x" % oracle_lib.fbhash(fb) == known["packet_frames"][1]["fbhash"]
    assert np.array_equal((rgb[..., 0] << 16) | (rgb[..., 1] << 8) | rgb[..., 2], fb & 0xFFFFFF)


def test_cpp_host_driver_moves_the_first_sphere(tmp_path):
    """The keys U and I (game.cpp:71-72) replayed by the driver's --script with the two spheres of raytracer.cpp:404-412
    traced (--spheres): the last frame equals the frame of a context whose first sphere was moved by the same float32 steps."""
    import os
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "raytracing_amd", "host", "mrt_game")
    out = str(tmp_path / "frame.ppm")
    dt = np.float32(0.05)
    subprocess.check_call([exe, "--scene", os.path.join(ROOT, "tests", "golden", "maze.mrts"), "--size", "512", "264", "--mode", "recursive",
                           "--spheres", "--script", "U*6 I*1", "--dt", repr(float(dt)), "--pos", "0", "0.3", "2.5", "--lookat", "0", "0", "0", "--out", out])
    with open(out, "rb") as f:
        assert f.readline() == b"P6\n" and f.readline() == b"512 264\n" and f.readline() == b"255\n"
        rgb = np.frombuffer(f.read(), np.uint8).reshape(264, 512, 3).astype(np.uint32)
    y = np.float32(0)
    for step in (1, 1, 1, 1, 1, 1, -1):
        y = np.float32(y + dt) if step > 0 else np.float32(y - dt)
    rt = Raytracer(512, 264, device=0)
    rt.build_bvh(scene_io.load_scene(os.path.join(ROOT, "tests", "golden", "maze.mrts")))
    frames = []
    for sy in (y, np.float32(0)):
        rt.set_objects(np.float32([[0, 1, sy, 0, 0.5, 1, 0, 1, 0, 0.5, 0.0, 1.0], [0, -1, 0, 0, 0.5, 1, 1, 1, 1, 0.0, 0.5, 1.5]]))
        frames.append(rt.render(oracle_lib.lookat((0, 0.3, 2.5), (0, 0, 0)), mode=MODE_RECURSIVE)[0])
    rt.close()
    got = (rgb[..., 0] << 16) | (rgb[..., 1] << 8) | rgb[..., 2]
    assert np.array_equal(got, frames[0] & 0xFFFFFF)
    assert not np.array_equal(frames[0], frames[1])          # (the sphere is in view: moving it changes the frame)


@pytest.mark.parametrize("args", [["--rehearse-ranks", "3", "--size", "512", "264", "--mode", "recursive", "--block", "2"],
                                  ["--rehearse-ranks", "8", "--size", "960", "544", "--mode", "packet", "--spp-grid", "4"]],
                         ids=["3 ranks recursive", "8 ranks 16 spp"])
def test_cpp_host_driver_split_render_with_native_gather(args):
    """mrt_game playing every rank of the multi-GPU split in one process: shares rendered packed, moved into rank 0's
    receive buffers and scattered by libmrt_gather.so; the driver itself compares with the undivided frame."""
    import os
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "raytracing_amd", "host", "mrt_game")
    r = subprocess.run([exe, "--scene", os.path.join(ROOT, "tests", "golden", "maze.mrts"), "--frames", "3", "--pos", "6", "3", "12",
                        "--lookat", "0", "0", "0"] + args, capture_output=True, text=True)
    assert r.returncode == 0 and "is identical to the undivided frame" in r.stdout, r.stdout[-400:] + r.stderr[-400:]


def test_cpp_host_driver_overlay_and_png_sequence(tmp_path, known):
    """mrt_game --overlay --out-sequence: every frame lands as a PNG; outside the two overlay lines the pixels are the
    reference frame's (800x600, default camera, packet path: the recorded hash), inside them some are the overlay's
    magenta."""
    import os
    import struct
    import subprocess
    import zlib
    from conftest import ROOT
    exe = os.path.join(ROOT, "raytracing_amd", "host", "mrt_game")
    prefix = str(tmp_path / "seq")
    subprocess.check_call([exe, "--scene", os.path.join(ROOT, "tests", "golden", "maze.mrts"), "--size", "800", "600", "--mode", "packet",
                           "--frames", "3", "--overlay", "--out-sequence", prefix])
    rt = Raytracer(800, 600, device=0)
    rt.build_bvh(scene_io.load_scene(os.path.join(ROOT, "tests", "golden", "maze.mrts")))
    fb, _ = rt.render(oracle_lib.IDENTITY, mode=MODE_PACKET)
    rt.close()
    assert "%016x" % oracle_lib.fbhash(fb) == known["packet_frames"][1]["fbhash"]
    for k in range(3):
        data = open("%s_%04d.png" % (prefix, k), "rb").read()
        assert data[:8] == b"\x89PNG\r\n\x1a\n" and struct.unpack(">II", data[16:24]) == (800, 600)
        n = struct.unpack(">I", data[33:37])[0]
        raw = np.frombuffer(zlib.decompress(data[41:41 + n]), np.uint8).reshape(600, 1 + 3 * 800)[:, 1:].reshape(600, 800, 3).astype(np.uint32)
        px = (raw[..., 0] << 16) | (raw[..., 1] << 8) | raw[..., 2]
        outside = np.ones((600, 800), bool)
        outside[2:17, 2:2 + 6 * 24] = False
        assert np.array_equal(px[outside], (fb & 0xFFFFFF)[outside])
        assert (px[2:17, 2:2 + 6 * 24] == 0xFF00FF).sum() > 100


def test_random_pose_campaign():
    """scripts/parity_fuzz.py: 80 random cameras (some on integer coordinates, some looking straight down), random
    light positions, four frame sizes, every fifth pose with a 2x2 sample grid, both shaders."""
    import subprocess
    import sys
    from conftest import ROOT
    import os
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "parity_fuzz.py"), "80", "3"], capture_output=True, text=True)
    assert r.returncode == 0 and "hit ids identical" in r.stdout, r.stdout[-500:] + r.stderr[-500:]
    # and 25 random triangle soups with random reflecting / refracting materials and 0-3 lights
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "parity_fuzz.py"), "--soups", "25", "9"], capture_output=True, text=True)
    assert r.returncode == 0 and "hit ids identical" in r.stdout, r.stdout[-500:] + r.stderr[-500:]


def test_api_sequence_fuzz():
    """scripts/api_fuzz.py: random sequences of calls on up to three contexts (scene uploads, light changes, both
    shaders, bands, interleaved shares, queued frames, rasterizer frames, destruction); every frame must equal the
    frame a fresh context renders, so no state (tile order, cost history, pair buffer, queue slots) leaks between calls."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "api_fuzz.py"), "250", "11"], capture_output=True, text=True)
    assert r.returncode == 0 and "all identical" in r.stdout, r.stdout[-500:] + r.stderr[-800:]
