// Headless stand-in for the reference's application loop (game.cpp:15-35 Init, :84-113 Tick):
// loads the scene, builds the BVH, renders frames with the ray tracer or the rasterizer and
// reports frame time and ray throughput the way the reference's overlay does (game.cpp:95-112).
//
//   mrt_game [--scene assets/maze.obj | scene.mrts] [--size W H] [--frames N]
//            [--mode packet|recursive|depth|raster] [--spheres] [--one-tree] [--pos x y z --lookat x y z]
//            [--out frame.ppm|frame.png] [--out-sequence prefix] [--overlay] [--dump-scene out.mrts] [--device N]
//            [--script "W*30 RIGHT*10 O*5 U*3 SPACE*1" --dt 0.016] [--no-render] [--pipeline]
//            [--ranks N --rank R --nccl-id FILE [--block B] [--spp-grid G]]  |  [--rehearse-ranks N]
//
// --script replays Game::HandleInput (game.cpp:41-75) without a keyboard: each token holds one key for N
// frames of length dt (W S A D R F move, LEFT RIGHT UP DOWN turn, O P move the light, SPACE switches
// between ray tracer and rasterizer).  --no-render runs the script only and prints the final state.
// --mode depth is the BVH-depth view (raytracer.cpp:581-586; the script key T toggles it like game.cpp:73);
// --spheres adds the two spheres Raytracer::Init creates (raytracer.cpp:404-412), which the reference never traces.
// --one-tree uploads the scene with mrt_scene_upload (the reference's tree alone) instead of mrt_scene_upload_lbvh_exact.
// --overlay draws the reference's frame-time / FPS lines into every frame (game.cpp:95-112), updated once per second of
// frame time like there; --out-sequence writes every frame as prefix_0000.png, prefix_0001.png, ... (the stand-in for the
// window the reference presents its Surface in, template.cpp:111-127).
// --pipeline keeps the next frame in flight while the finished one is presented (two Surfaces,
// mrt_render_submit / mrt_render_wait); ray-traced frames only.
// --ranks N --rank R: one process per GPU (GPU R unless --device says otherwise).  The scene is replicated,
// every process renders its interleaved share of the frame's tile rows (mrt_render_device_part) and the shares
// are gathered on rank 0 with RCCL (mrt_gather_exchange); rank 0 reports and writes the frame.  The ranks find
// each other through FILE: rank 0 writes the communicator's id there, the others wait for it.
// --rehearse-ranks N: the same split and gather in ONE process on one GPU, a device copy standing in for the
// transport; checks the assembled frame against the undivided one.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sstream>
#include <string>
#include <vector>

#include <hip/hip_runtime_api.h>
#include <thread>

#include "../../include/mrt_gather.h"
#include "mrt_host.hpp"

using namespace mrt_host;

namespace {
struct MultiOpts { int ranks = 1, rank = 0, block = 1, grid = 1, rehearse = 0; std::string id_file; };

#define HIP_OK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) throw std::runtime_error(std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)
#define MRT_CHECK(ctx, call) do { int rc_ = (call); if (rc_ != MRT_OK) throw std::runtime_error(std::string(#call) + ": " + mrt_last_error(ctx)); } while (0)
#define GATHER_OK(g, call) do { int rc_ = (call); if (rc_ != MRT_OK) throw std::runtime_error(std::string(#call) + ": " + mrt_gather_last_error(g)); } while (0)

// rank 0 writes the id under a temporary name and renames it, so a reader never sees half a file
struct CommId { unsigned char bytes[MRT_GATHER_ID_BYTES]; };
CommId share_id(const MultiOpts& m) {
    CommId id;
    if (m.rank == 0) {
        if (mrt_gather_comm_id(id.bytes) != MRT_OK) throw std::runtime_error(std::string("mrt_gather_comm_id: ") + mrt_gather_last_error(nullptr));
        const std::string tmp = m.id_file + ".tmp";
        FILE* f = fopen(tmp.c_str(), "wb");
        if (!f || fwrite(&id, sizeof id, 1, f) != 1) throw std::runtime_error("cannot write " + tmp);
        fclose(f);
        if (rename(tmp.c_str(), m.id_file.c_str()) != 0) throw std::runtime_error("cannot rename to " + m.id_file);
        return id;
    }
    for (int tries = 0; tries < 3000; ++tries) {          // up to five minutes
        FILE* f = fopen(m.id_file.c_str(), "rb");
        if (f) {
            const bool ok = fread(&id, sizeof id, 1, f) == 1;
            fclose(f);
            if (ok) return id;
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(100));
    }
    throw std::runtime_error("no communicator id appeared in " + m.id_file);
}

// One frame loop of the split render.  Real ranks: this process is rank m.rank of m.ranks.  Rehearsal: this
// process plays every rank in turn on its one GPU.
int run_split(const MultiOpts& m, mrt_host::Scene& scene, int W, int H, int frames, const std::string& mode, const mrt_host::Camera& camera,
              int device, const std::string& out) {
    using namespace mrt_host;
    const bool rehearsal = m.rehearse > 0;
    const int n = rehearsal ? m.rehearse : m.ranks, me = rehearsal ? 0 : m.rank;
    const uint32_t flags = (mode == "recursive" ? MRT_MODE_RECURSIVE : MRT_MODE_PACKET) | MRT_SPP_GRID(m.grid);
    HIP_OK(hipSetDevice(device >= 0 ? device : (rehearsal ? 0 : m.rank)));
    void* comm = nullptr;
    if (!rehearsal) {
        const CommId id = share_id(m);
        if (mrt_gather_comm_create(id.bytes, me, n, &comm) != MRT_OK) throw std::runtime_error(std::string("mrt_gather_comm_create: ") + mrt_gather_last_error(nullptr));
    }
    Surface screen(W, H);
    Raytracer raytracer;
    raytracer.Init(&screen, -1);
    raytracer.BuildBVH(scene);
    mrt_ctx* ctx = raytracer.Context();
    mrt_gather* g = nullptr;
    if (mrt_gather_create(W, H, me, n, m.block, &g) != MRT_OK) throw std::runtime_error(std::string("mrt_gather_create: ") + mrt_gather_last_error(nullptr));
    hipStream_t stream;
    HIP_OK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    const int my_rows = me == 0 ? H : mrt_gather_share_rows(g, me);
    uint32_t *fb = nullptr, *packed = nullptr;
    HIP_OK(hipMalloc((void**)&fb, (size_t)(my_rows > 0 ? my_rows : 8) * W * 4));
    if (rehearsal) HIP_OK(hipMalloc((void**)&packed, (size_t)H * W * 4));
    double total_ms = 0, best_ms = 1e30;
    for (int f = 0; f < frames; ++f) {
        HIP_OK(hipStreamSynchronize(stream));
        const auto t0 = std::chrono::steady_clock::now();
        MRT_CHECK(ctx, mrt_render_device_part(ctx, camera.transform, flags, me, n, m.block, me != 0, fb, W, nullptr, stream));
        if (rehearsal) {
            for (int r = 1; r < n; ++r) {
                if (mrt_gather_share_rows(g, r) == 0) continue;
                MRT_CHECK(ctx, mrt_render_device_part(ctx, camera.transform, flags, r, n, m.block, 1, packed, W, nullptr, stream));
                GATHER_OK(g, mrt_gather_stage_local(g, r, packed, stream));
                GATHER_OK(g, mrt_gather_scatter(g, r, fb, W, stream));
            }
        } else {
            GATHER_OK(g, mrt_gather_exchange(g, comm, fb, W, stream));
        }
        if (me == 0) HIP_OK(hipMemcpyAsync(screen.GetBuffer(), fb, (size_t)H * W * 4, hipMemcpyDeviceToHost, stream));
        HIP_OK(hipStreamSynchronize(stream));
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (f > 0 || frames == 1) { total_ms += ms; best_ms = ms < best_ms ? ms : best_ms; }
    }
    int rc = 0;
    if (me == 0) {
        const int counted = frames > 1 ? frames - 1 : 1;
        printf("%s of %d: Frametime : %f ms (best %f), FPS : %.1f  [%dx%d, %d x %d samples, %s, tile rows dealt in blocks of %d]\n",
               rehearsal ? "rehearsal" : "rank 0", n, total_ms / counted, best_ms, 1000.0 * counted / total_ms, W, H, m.grid, m.grid, mode.c_str(), m.block);
        if (rehearsal) {
            std::vector<uint32_t> whole((size_t)W * H);
            MRT_CHECK(ctx, mrt_render_device(ctx, camera.transform, flags, 0, H, packed, W, nullptr, stream));
            HIP_OK(hipMemcpyAsync(whole.data(), packed, whole.size() * 4, hipMemcpyDeviceToHost, stream));
            HIP_OK(hipStreamSynchronize(stream));
            const bool same = !memcmp(whole.data(), screen.GetBuffer(), whole.size() * 4);
            printf("frame assembled from %d shares %s the undivided frame\n", n, same ? "is identical to" : "DIFFERS from");
            if (!same) rc = 3;
        }
        if (!out.empty() && screen.SavePPM(out)) printf("wrote %s\n", out.c_str());
    }
    mrt_gather_comm_destroy(comm);
    mrt_gather_destroy(g);
    (void)hipFree(fb);
    if (packed) (void)hipFree(packed);
    (void)hipStreamDestroy(stream);
    return rc;
}
}  // namespace

int main(int argc, char** argv) {
    MultiOpts multi;
    std::string scene_file = "assets/maze.obj", mode = "packet", out, dump, script;
    float dt = 0.016f;
    bool no_render = false, pipeline = false, overlay = false, spheres = false, one_tree = false;
    std::string sequence;
    int W = 800, H = 600, frames = 10, device = -1;       // game.h:3-4
    bool have_pose = false;
    vec3 pos{0, 0, 0}, target{0, 0, -1};
    for (int i = 1; i < argc; ++i) {
        auto need = [&](int n) { if (i + n >= argc) { fprintf(stderr, "missing value after %s\n", argv[i]); exit(2); } };
        if (!strcmp(argv[i], "--scene")) { need(1); scene_file = argv[++i]; }
        else if (!strcmp(argv[i], "--size")) { need(2); W = atoi(argv[++i]); H = atoi(argv[++i]); }
        else if (!strcmp(argv[i], "--frames")) { need(1); frames = atoi(argv[++i]); }
        else if (!strcmp(argv[i], "--mode")) { need(1); mode = argv[++i]; }
        else if (!strcmp(argv[i], "--out")) { need(1); out = argv[++i]; }
        else if (!strcmp(argv[i], "--dump-scene")) { need(1); dump = argv[++i]; }
        else if (!strcmp(argv[i], "--device")) { need(1); device = atoi(argv[++i]); }
        else if (!strcmp(argv[i], "--script")) { need(1); script = argv[++i]; }
        else if (!strcmp(argv[i], "--dt")) { need(1); dt = (float)atof(argv[++i]); }
        else if (!strcmp(argv[i], "--no-render")) no_render = true;
        else if (!strcmp(argv[i], "--pipeline")) pipeline = true;
        else if (!strcmp(argv[i], "--overlay")) overlay = true;
        else if (!strcmp(argv[i], "--spheres")) spheres = true;
        else if (!strcmp(argv[i], "--one-tree")) one_tree = true;       // mrt_scene_upload instead of mrt_scene_upload_lbvh_exact
        else if (!strcmp(argv[i], "--out-sequence")) { need(1); sequence = argv[++i]; }
        else if (!strcmp(argv[i], "--ranks")) { need(1); multi.ranks = atoi(argv[++i]); }
        else if (!strcmp(argv[i], "--rank")) { need(1); multi.rank = atoi(argv[++i]); }
        else if (!strcmp(argv[i], "--nccl-id")) { need(1); multi.id_file = argv[++i]; }
        else if (!strcmp(argv[i], "--block")) { need(1); multi.block = atoi(argv[++i]); }
        else if (!strcmp(argv[i], "--spp-grid")) { need(1); multi.grid = atoi(argv[++i]); }
        else if (!strcmp(argv[i], "--rehearse-ranks")) { need(1); multi.rehearse = atoi(argv[++i]); }
        else if (!strcmp(argv[i], "--pos")) { need(3); pos = {(float)atof(argv[i + 1]), (float)atof(argv[i + 2]), (float)atof(argv[i + 3])}; i += 3; have_pose = true; }
        else if (!strcmp(argv[i], "--lookat")) { need(3); target = {(float)atof(argv[i + 1]), (float)atof(argv[i + 2]), (float)atof(argv[i + 3])}; i += 3; have_pose = true; }
        else { fprintf(stderr, "unknown option %s\n", argv[i]); return 2; }
    }
    try {
        Scene scene;
        scene.Add(scene_file);                                  // game.cpp:19
        if (scene_file.size() < 5 || scene_file.substr(scene_file.size() - 5) != ".mrts") scene.ApplyReferenceMaterials();
        size_t tris = 0;
        for (const Mesh& m : scene.meshList) tris += m.tris();
        printf("scene %s: %zu meshes, %zu triangles, %zu textures\n", scene_file.c_str(), scene.meshList.size(), tris, scene.texList.size());
        if (!dump.empty()) { scene.SaveMRTS(dump); printf("wrote %s\n", dump.c_str()); return 0; }

        // one entry per frame: the key held during that frame ("" = none)
        std::vector<std::string> keys;
        {
            std::istringstream in(script);
            std::string tok;
            while (in >> tok) {
                const size_t star = tok.find('*');
                const int n = star == std::string::npos ? 1 : atoi(tok.c_str() + star + 1);
                for (int k = 0; k < n; ++k) keys.push_back(tok.substr(0, star));
            }
        }
        if (!keys.empty()) frames = (int)keys.size();

        if (multi.ranks > 1 || multi.rehearse > 0) {
            if (multi.rehearse <= 0 && (multi.id_file.empty() || multi.rank < 0 || multi.rank >= multi.ranks)) {
                fprintf(stderr, "--ranks needs --rank R (0 <= R < N) and --nccl-id FILE\n");
                return 2;
            }
            Camera cam;
            if (have_pose) { cam.SetPosition(pos); cam.LookAt(target); }
            return run_split(multi, scene, W, H, frames, mode, cam, device, out);
        }
        Surface screen(W, H);                                   // template.cpp:98
        Surface screen2(pipeline ? W : 8, pipeline ? H : 8);    // second frame of the pipelined loop
        int prev_ticket = -1;
        unsigned long long shown_sum = 0;
        Camera camera;
        vec3 position = {0, 0, 0};                              // game.cpp:4, 33
        if (have_pose) { camera.SetPosition(pos); camera.LookAt(target); position = pos; }
        std::vector<float> light = {0, 5, 0, 1, 1, 1, 1};       // raytracer.cpp:396-399
        bool is_raytracer = mode != "raster", depth_view = mode == "depth";
        // the two spheres Raytracer::Init creates (raytracer.cpp:404-412): the keys U and I move the first one
        // (game.cpp:71-72) whether or not it is traced (--spheres)
        std::vector<mrt_object> objects = {mrt_object{MRT_OBJECT_SPHERE, {1, 0, 0}, 0.5f, {1, 0, 1, 0}, 0.5f, 0.0f, 1.0f},
                                           mrt_object{MRT_OBJECT_SPHERE, {-1, 0, 0}, 0.5f, {1, 1, 1, 1}, 0.0f, 0.5f, 1.5f}};
        Raytracer raytracer;
        auto t0 = std::chrono::steady_clock::now();
        if (!no_render) {
            raytracer.Init(&screen, device);                    // game.cpp:17-18
            raytracer.BuildBVH(scene, !one_tree);               // game.cpp:34
            if (spheres) raytracer.SetObjects(objects);
            printf("BuildBVH + upload: %.2f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        }
        auto add = [](vec3 a, vec3 b) { return vec3{a.x + b.x, a.y + b.y, a.z + b.z}; };
        auto sub = [](vec3 a, vec3 b) { return vec3{a.x - b.x, a.y - b.y, a.z - b.z}; };
        auto mul = [](vec3 a, float k) { return vec3{a.x * k, a.y * k, a.z * k}; };

        double total_ms = 0, best_ms = 1e30;
        uint64_t rays[4] = {0, 0, 0, 0};
        // Game::Tick's counters (game.cpp:95-105): frames and their time are added up, and once a second of frame time has
        // passed the displayed values are refreshed
        int tick_frames = 0, frames_per_second = 0;
        float tick_time = 0.0f, frame_duration = 0.0f;
        auto present = [&](Surface& shown, int frame_no, float frame_dt) {
            tick_frames++;
            tick_time += frame_dt;
            if (tick_time >= 1.0f) {
                frames_per_second = tick_frames;
                frame_duration = tick_time / (float)tick_frames;
                tick_frames = 0;
                tick_time = 0.0f;
            }
            if (overlay) shown.PrintFrameStats(frame_duration, frames_per_second);
            if (!sequence.empty()) {
                char name[32];
                snprintf(name, sizeof name, "_%04d.png", frame_no);
                if (!shown.SavePNG(sequence + name)) throw std::runtime_error("cannot write " + sequence + name);
            }
        };
        for (int f = 0; f < frames; ++f) {                      // Game::Tick
            if (!keys.empty()) {                                // Game::HandleInput, game.cpp:41-75
                const std::string& k = keys[f];
                if (k == "SPACE") is_raytracer = !is_raytracer;
                if (k == "T") depth_view = !depth_view;         // game.cpp:73
                if (k == "W") position = add(position, mul(mul(camera.GetForward(), dt), 10.0f));
                if (k == "S") position = sub(position, mul(mul(camera.GetForward(), dt), 10.0f));
                if (k == "D") position = add(position, mul(mul(camera.GetRight(), dt), 10.0f));
                if (k == "A") position = sub(position, mul(mul(camera.GetRight(), dt), 10.0f));
                if (k == "R") position = add(position, mul(mul(camera.GetUp(), dt), 10.0f));
                if (k == "F") position = sub(position, mul(mul(camera.GetUp(), dt), 10.0f));
                camera.SetPosition(position);
                const vec3 ahead = add(camera.GetPosition(), camera.GetForward());
                if (k == "RIGHT") camera.LookAt(add(ahead, mul(camera.GetRight(), dt * 2.0f)));
                if (k == "LEFT") camera.LookAt(sub(ahead, mul(camera.GetRight(), dt * 2.0f)));
                if (k == "DOWN") camera.LookAt(add(ahead, mul(camera.GetUp(), dt * 2.0f)));
                if (k == "UP") camera.LookAt(sub(ahead, mul(camera.GetUp(), dt * 2.0f)));
                if (k == "O") light[1] += dt;
                if (k == "P") light[1] -= dt;
                if (!no_render && (k == "O" || k == "P")) raytracer.SetLights(light);
                if (k == "U") objects[0].pos[1] += dt;          // game.cpp:71
                if (k == "I") objects[0].pos[1] -= dt;          // game.cpp:72
                if (!no_render && spheres && (k == "U" || k == "I")) raytracer.SetObjects(objects);
            }
            if (no_render) continue;
            if (pipeline && is_raytracer) {
                // frame f renders while frame f-1 is presented (here: folded into a checksum)
                Surface* target = f % 2 ? &screen2 : &screen;
                const auto ts = std::chrono::steady_clock::now();
                const int ticket = raytracer.Submit(camera, target, mode == "recursive");
                if (f > 0) {
                    raytracer.Wait(prev_ticket);
                    Surface* shown = f % 2 ? &screen : &screen2;
                    for (int i = 0; i < W * H; i += 4099) shown_sum += shown->GetBuffer()[i];
                    present(*shown, f - 1, keys.empty() ? (float)(total_ms / (f > 1 ? f - 1 : 1) * 1e-3) : dt);
                }
                prev_ticket = ticket;
                if (f == frames - 1) {
                    raytracer.Wait(ticket);
                    if (target != &screen) memcpy(screen.GetBuffer(), target->GetBuffer(), (size_t)W * H * sizeof(uint32_t));
                    present(screen, f, keys.empty() ? (float)(total_ms / (f > 1 ? f - 1 : 1) * 1e-3) : dt);
                }
                const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ts).count();
                if (f > 0 || frames == 1) { total_ms += ms; best_ms = ms < best_ms ? ms : best_ms; }
                continue;
            }
            screen.Clear(0);                                    // game.cpp:87
            t0 = std::chrono::steady_clock::now();
            if (!is_raytracer) raytracer.RenderRaster(camera);
            else if (depth_view) raytracer.RenderDepth(camera);
            else if (mode == "recursive") raytracer.RenderRecursive(camera);
            else if (mode == "packet" || mode == "raster" || mode == "depth") raytracer.Render(camera);
            else { fprintf(stderr, "unknown mode %s\n", mode.c_str()); return 2; }
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (f > 0 || frames == 1) { total_ms += ms; best_ms = ms < best_ms ? ms : best_ms; }
            present(screen, f, keys.empty() ? (float)(ms * 1e-3) : dt);      // a replayed script advances by its own dt, as Tick(dt) does
        }
        if (!keys.empty()) {
            printf("camera:");
            for (int i = 0; i < 16; ++i) { uint32_t u; memcpy(&u, &camera.transform[i], 4); printf(" %08x", u); }
            uint32_t lu, ou;
            memcpy(&lu, &light[1], 4);
            memcpy(&ou, &objects[0].pos[1], 4);
            printf("\nlight.y: %08x  object0.y: %08x  renderer: %s\n", lu, ou, is_raytracer ? "raytracer" : "rasterizer");
        }
        if (no_render) return 0;
        const int counted = frames > 1 ? frames - 1 : 1;
        if (is_raytracer) raytracer.RayCounts(rays);
        const double n_rays = (double)(rays[0] + rays[1] + rays[2]);
        printf("Frametime : %f ms (best %f), FPS : %.1f", total_ms / counted, best_ms, 1000.0 * counted / total_ms);
        // pipelined: one iteration is not one frame's latency, only the average says how often a frame lands
        const double per_frame_ms = pipeline ? total_ms / counted : best_ms;
        if (n_rays > 0) printf(", %.0f rays/frame, %.1f Mrays/s (host framebuffer copy included)", n_rays, n_rays / per_frame_ms / 1e3);
        if (pipeline) printf(" [pipelined, checksum %llx]", shown_sum);
        printf("\n");
        const bool png = out.size() > 4 && out.substr(out.size() - 4) == ".png";
        if (!out.empty() && (png ? screen.SavePNG(out) : screen.SavePPM(out))) printf("wrote %s\n", out.c_str());
    } catch (const std::exception& e) {
        fprintf(stderr, "mrt_game: %s\n", e.what());
        return 1;
    }
    return 0;
}
