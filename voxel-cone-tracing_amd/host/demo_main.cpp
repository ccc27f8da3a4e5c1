// demo_main.cpp -- headless caller with the reference application's call sequence
// (R/main.cpp:64-68 set-up, :77-94 frame loop) against the facade header: proves that a caller
// written for the reference's orchestrator drives the MI355X path unchanged.  Instead of a GLFW
// window and swap-buffers it renders N frames, prints a checksum of the last RGBA16F frame and
// optionally writes it as a tonemapped PPM.
//
//   vct_demo [--scene procedural:atrium|procedural:atrium-textured|procedural:bistro|procedural:cornell] [--voxels 128] [--size 1280x720]
//            [--shadow 4096] [--frames 3] [--bounces 1|2] [--ppm out.ppm] [--gpus N] [--dynamic-light] [--frames-in-flight 1|2]
//
// --frames-in-flight 2: consecutive Render() calls alternate between two frame slots (Voxel_Cone_Tracing::FramesInFlight):
//   frame k + 1 starts while frame k drains; same pixels, same checksum.
// --dynamic-light: every Render() re-runs the whole GI pass (shadow map, voxelize, inject, mips, G-buffer, trace)
// for the current lightDirection through vct_gi_pass instead of the reference's build-once volume.
//
// --gpus N: the frame is cut into N screen-tile slabs, one process per GPU (this program re-launches
// itself N times BEFORE anything touches a GPU; rank r uses device r), each rank rasterises and traces its
// slab, rank 0 receives the frame through one ncclGather per frame and prints / writes it.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <string>
#include <vector>

#include "Voxel_Cone_Tracing.h"

static const int SCREEN_WIDTH = 1280;
static const int SCREEN_HEIGHT = 720;

static float half_to_float(uint16_t h) {
    const uint32_t s = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
    float f;
    if (e == 0) f = ldexpf((float)m, -24);
    else if (e == 31) f = m ? NAN : INFINITY;
    else f = ldexpf((float)(m | 0x400u), (int)e - 25);
    return s ? -f : f;
}

// parent of a multi-GPU run: start one child per rank (fork + exec of this binary; the parent never
// initialises a GPU) and wait for them
static int launch_ranks(int gpus, int argc, char** argv) {
    char idfile[64];
    snprintf(idfile, sizeof(idfile), "/tmp/vct_demo_id_%d", (int)getpid());
    unlink(idfile);
    std::vector<pid_t> kids;
    for (int r = 0; r < gpus; ++r) {
        pid_t pid = fork();
        if (pid == 0) {
            std::vector<std::string> a(argv, argv + argc);
            a.push_back("--rank"); a.push_back(std::to_string(r));
            a.push_back("--idfile"); a.push_back(idfile);
            std::vector<char*> av;
            for (auto& x : a) av.push_back(const_cast<char*>(x.c_str()));
            av.push_back(nullptr);
            execv("/proc/self/exe", av.data());
            perror("execv");
            _exit(127);
        }
        kids.push_back(pid);
    }
    int rc = 0;
    for (pid_t k : kids) {
        int st = 0;
        waitpid(k, &st, 0);
        if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) rc = 5;
    }
    unlink(idfile);
    return rc;
}

int main(int argc, char** argv) {
    int w = SCREEN_WIDTH, h = SCREEN_HEIGHT, frames = 3, voxels = 128, shadow = 4096, bounces = 1;
    int gpus = 0, rank = -1, in_flight = 1;
    const char* scene = "procedural:atrium";
    const char* ppm = nullptr;
    bool dynamic_light = false;
    const char* idfile = nullptr;
    for (int i = 1; i + 1 < argc; ++i) {
        if (!strcmp(argv[i], "--gpus")) gpus = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--rank")) rank = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--idfile")) idfile = argv[++i];
    }
    if (gpus > 0 && rank < 0) return launch_ranks(gpus, argc, argv);
    // (options with a value consume it; --dynamic-light has none and may stand anywhere)
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--dynamic-light")) { dynamic_light = true; continue; }
        if (i + 1 >= argc) break;
        if (!strcmp(argv[i], "--scene")) scene = argv[++i];
        else if (!strcmp(argv[i], "--voxels")) voxels = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--size")) sscanf(argv[++i], "%dx%d", &w, &h);
        else if (!strcmp(argv[i], "--shadow")) shadow = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--frames")) frames = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--bounces")) bounces = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--ppm")) ppm = argv[++i];
        else if (!strcmp(argv[i], "--frames-in-flight")) in_flight = atoi(argv[++i]);
    }
    GLFWwindow* window = nullptr;          // no window system on a compute node

    camera.MovementSpeed = 5.0f;           // R/main.cpp:64-65
    camera.MouseSensitivity = 0.5f;
    if (!strcmp(scene, "procedural:cornell")) {
        camera.position = vec3(0.0f, 0.0f, 58.0f);
    } else {
        camera.position = vec3(-56.0f, -9.0f, 2.0f);
        camera.Yaw = 0.0f; camera.Pitch = 8.0f;
        camera.UpdateCamera();
    }
    Voxel_Cone_Tracing voxel_cone_tracing(w, h, window);    // R/main.cpp:66
    voxel_cone_tracing.VoxelDimensions = voxels;
    voxel_cone_tracing.ShadowMapSize = (unsigned)shadow;
    voxel_cone_tracing.model_path = scene;
    voxel_cone_tracing.Bounces = bounces;
    voxel_cone_tracing.DynamicLight = dynamic_light;       // every Render() = one whole GI pass (vct_gi_pass)
    voxel_cone_tracing.FramesInFlight = in_flight;
    if (gpus > 0) {                                         // a rank of a multi-GPU run
        voxel_cone_tracing.Rank = rank;
        voxel_cone_tracing.World = gpus;
        // (VCT_DEMO_SINGLE_DEVICE=1: every rank on device 0 -- only the direct-slab mode, VCT_COMM_MODE=direct, accepts that;
        // it is how that mode is tested on a one-GPU box)
        voxel_cone_tracing.Device = getenv("VCT_DEMO_SINGLE_DEVICE") ? 0 : rank;
        const std::string tmp = std::string(idfile) + ".tmp";
        if (rank == 0) {                                    // create the RCCL id, publish it atomically
            if (vct_comm_get_unique_id(voxel_cone_tracing.CommId) != VCT_OK) { printf("%s\n", vct_last_error(nullptr)); return 6; }
            FILE* fp = fopen(tmp.c_str(), "wb");
            if (!fp || fwrite(voxel_cone_tracing.CommId, 1, VCT_COMM_ID_BYTES, fp) != VCT_COMM_ID_BYTES) return 6;
            fclose(fp);
            rename(tmp.c_str(), idfile);
        } else {
            FILE* fp = nullptr;
            for (int tries = 0; tries < 6000 && !(fp = fopen(idfile, "rb")); ++tries) usleep(10000);
            if (!fp || fread(voxel_cone_tracing.CommId, 1, VCT_COMM_ID_BYTES, fp) != VCT_COMM_ID_BYTES) return 6;
            fclose(fp);
        }
    }
    voxel_cone_tracing.init_voxel_cone_tracing();           // R/main.cpp:68
    if (voxel_cone_tracing.last_status != VCT_OK) return 2;

    float delta_time = 0.05f;
    voxel_cone_tracing.Render();                            // frame 0 pays first-launch costs
    voxel_cone_tracing.Finish();
    if (voxel_cone_tracing.last_status != VCT_OK) return 3;
    // R/main.cpp:77-94 as written: Render() per frame, nothing read back (the reference swaps buffers instead)
    auto t0 = std::chrono::steady_clock::now();
    for (int f = 1; f < frames; ++f) {
        camera.ProcessKeyBoard(FORWARD, delta_time);
        voxel_cone_tracing.Render();
        if (voxel_cone_tracing.last_status != VCT_OK) return 3;
    }
    voxel_cone_tracing.Finish();
    const double wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    // the same number of frames with the RGBA16F frame copied to the host after every Render() (a presenter without
    // GPU interop): what the per-frame download costs
    double readback_ms = 0.0;
    if (frames > 1 && gpus <= 0) {
        const vec3 keep = camera.position;
        auto t1 = std::chrono::steady_clock::now();
        for (int f = 1; f < frames; ++f) {
            voxel_cone_tracing.Render();
            if (!voxel_cone_tracing.Frame()) return 3;
        }
        readback_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
        camera.position = keep;
    }
    if (gpus > 0 && rank != 0) { voxel_cone_tracing.Frame(); return 0; }       // the frame lives on rank 0
    if (gpus > 0) printf("gpus=%d (screen-tile slabs + one ncclGather per frame)\n", gpus);
    if (frames > 1) {
        printf("Render(): %.3f ms per frame (wall, %d frames, frame 0 excluded, no read-back, %d frame%s in flight)\n", wall_ms / (frames - 1), frames - 1,
               voxel_cone_tracing.FramesInFlight, voxel_cone_tracing.FramesInFlight == 1 ? "" : "s");
        if (gpus <= 0) printf("Render() + Frame(): %.3f ms per frame (frame copied to the host every frame)\n", readback_ms / (frames - 1));
    }
    // the facade issues its frames without timing events; the last frame once more with them, for the trace_ms below
    // (the camera has not moved since: the same frame)
    // (single GPU only: a rank's Render() is a collective step, and the other ranks are done)
    if (gpus <= 0) {
        vct_set_trace_timing(voxel_cone_tracing.ctx, 1);
        voxel_cone_tracing.Render();
        if (voxel_cone_tracing.last_status != VCT_OK) return 3;
    }
    const uint16_t* fr = voxel_cone_tracing.Frame();
    const size_t n = (size_t)w * h * 4;
    uint64_t sum = 1469598103934665603ull;                  // FNV-1a over the RGBA16F halves
    for (size_t i = 0; i < n; ++i) { sum ^= fr[i]; sum *= 1099511628211ull; }
    uint64_t steps = 0;
    vct_last_step_count(voxel_cone_tracing.ctx, &steps);
    float ms = 0.0f;
    if (gpus <= 0) vct_last_trace_ms(voxel_cone_tracing.ctx, &ms);      // (a rank's slab steps were not timed: 0)
    printf("frames=%d size=%dx%d voxels=%d cone_steps=%llu trace_ms=%.3f fnv1a=%016llx\n", frames, w, h,
           voxels, (unsigned long long)steps, ms, (unsigned long long)sum);
    if (ppm) {
        FILE* fp = fopen(ppm, "wb");
        if (!fp) return 4;
        fprintf(fp, "P6\n%d %d\n255\n", w, h);
        for (int y = h - 1; y >= 0; --y)            // row 0 is the bottom row of the GL window
            for (int x = 0; x < w; ++x) {
                unsigned char px[3];
                for (int c = 0; c < 3; ++c) {
                    float v = half_to_float(fr[((size_t)y * w + x) * 4 + c]);
                    v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
                    px[c] = (unsigned char)(powf(v, 1.0f / 2.2f) * 255.0f + 0.5f);
                }
                fwrite(px, 1, 3, fp);
            }
        fclose(fp);
    }
    return 0;
}
