// sanitize_check.cpp -- drives every entry point of the CPU oracle and of the host library once on
// small inputs; built with -fsanitize=address,undefined by `make -C oracle sanitize` (GPU ASan is
// not available on the pool, so memory errors are hunted on the CPU build).  Test infrastructure.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include <dirent.h>

#include <string>

#include "../voxel-cone-tracing_amd/host/vct_host.h"
#include "../voxel-cone-tracing_amd/host/vct_image.h"
#include "vct_oracle.h"

// The image decoders (PNG / inflate, JPEG, BMP, TGA, PNM) on every file of $VCT_SANITIZE_IMAGES and on thousands of
// corrupted copies of each (flipped bytes, truncations): whatever they return, they must not touch memory outside
// their buffers or trip UBSan.  tests/test_sanitize.py writes the sample files.
static int fuzz_images(const char* dir) {
    DIR* d = opendir(dir);
    if (!d) return 0;
    int files = 0, decoded = 0, variants = 0;
    uint64_t rng = 0x9e3779b97f4a7c15ull;
    auto next = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
    while (dirent* e = readdir(d)) {
        if (e->d_name[0] == '.') continue;
        const std::string path = std::string(dir) + "/" + e->d_name;
        FILE* fp = fopen(path.c_str(), "rb");
        if (!fp) continue;
        std::vector<uint8_t> buf;
        uint8_t tmp[4096];
        size_t n;
        while ((n = fread(tmp, 1, sizeof(tmp), fp)) > 0) buf.insert(buf.end(), tmp, tmp + n);
        fclose(fp);
        ++files;
        vct_image::Image im;
        if (vct_image::decode(buf, im)) ++decoded;
        for (int k = 0; k < 1500 && buf.size() > 16; ++k) {
            std::vector<uint8_t> b = buf;
            if (k % 5 == 4) b.resize((size_t)(next() % b.size()));
            else for (int j = 0, m = 1 + (int)(next() % 4); j < m; ++j) b[(size_t)(next() % b.size())] ^= (uint8_t)(1 + next() % 255);
            vct_image::Image v;
            (void)vct_image::decode(b, v);
            ++variants;
        }
    }
    closedir(d);
    printf("image decoders: %d files (%d decoded), %d corrupted variants\n", files, decoded, variants);
    return files > 0 && decoded == 0;      // sample files were given but none decoded: the check itself is broken
}

int main() {
    const int V = 16, w = 20, h = 12, S = 64;
    vcto_params p;
    vcto_default_params(&p);
    p.V = V;
    // scenes + host stages
    for (int kind = 0; kind < 3; ++kind) {      // Cornell, atrium, textured atrium
        vcth_scene* s = vcth_scene_create(kind, 0.05f, 7u);
        const int ntri = vcth_scene_num_triangles(s), nmat = vcth_scene_num_materials(s);
        std::vector<float> pos((size_t)ntri * 9), alb((size_t)nmat * 4), spec((size_t)nmat * 3);
        std::vector<float> nrm((size_t)ntri * 9), tan((size_t)ntri * 9), bit((size_t)ntri * 9);
        std::vector<int32_t> mat((size_t)ntri);
        vcth_scene_get(s, pos.data(), mat.data(), alb.data(), spec.data());
        vcth_scene_get_frames(s, nrm.data(), tan.data(), bit.data());
        std::vector<float> uv((size_t)ntri * 6);
        std::vector<int32_t> mat_tex((size_t)nmat * 3);
        vcth_scene_get_uvs(s, uv.data());
        vcth_scene_get_material_textures(s, mat_tex.data());
        const int ntex = vcth_scene_num_textures(s);
        std::vector<std::vector<uint8_t>> texels((size_t)ntex);
        std::vector<vcto_texture> tex((size_t)ntex);
        for (int i = 0; i < ntex; ++i) {
            int32_t tw, th;
            vcth_scene_texture_info(s, i, &tw, &th);
            texels[(size_t)i].resize((size_t)tw * th * 4);
            vcth_scene_get_texture(s, i, texels[(size_t)i].data());
            tex[(size_t)i] = {texels[(size_t)i].data(), tw, th};
        }
        vcto_mesh mesh;
        memset(&mesh, 0, sizeof(mesh));
        mesh.pos = pos.data(); mesh.nrm = nrm.data(); mesh.tan = tan.data(); mesh.bit = bit.data();
        mesh.uv = uv.data(); mesh.material = mat.data(); mesh.albedo = alb.data(); mesh.specular = spec.data();
        mesh.mat_tex = mat_tex.data(); mesh.textures = tex.data();
        mesh.ntri = ntri; mesh.nmat = nmat; mesh.ntex = ntex; mesh.model_scale = 0.05f;
        const float L[3] = {0.0f, 1.0f, 0.25f};
        float lvp[16], vp[16];
        vcth_light_view_proj(L, lvp);
        std::vector<float> depth((size_t)S * S), planes((size_t)23 * w * h);
        vcto_render_shadow_map(&mesh, lvp, S, depth.data());
        vcth_camera cam;
        vcth_default_camera(&cam);
        cam.position[2] = kind == 0 ? 58.0f : 2.0f;
        vcth_camera_view_proj(&cam, w, h, vp);
        vcto_render_gbuffer(&mesh, vp, w, h, depth.data(), S, lvp, planes.data());
        // oracle: voxelize (both modes, attributes), mips, aniso, bounce, trace
        vcto_scene sc;
        memset(&sc, 0, sizeof(sc));
        sc.pos = pos.data(); sc.material = mat.data(); sc.albedo = alb.data();
        sc.ntri = ntri; sc.nmat = nmat; sc.model_scale = 0.05f;
        sc.shadow_depth = depth.data(); sc.shadow_size = S;
        memcpy(sc.light_vp, lvp, sizeof(lvp));
        sc.uv = uv.data(); sc.mat_tex = mat_tex.data(); sc.textures = tex.data(); sc.ntex = ntex;
        const size_t nvox = (size_t)V * V * V, nchain = vcto_chain_texels(V);
        std::vector<uint8_t> l0(nvox * 4, 0), lref(nvox * 4, 0), a_alb(nvox * 4), a_nrm(nvox * 4), l1(nvox * 4);
        std::vector<uint32_t> acc(nvox * 4);
        vcto_voxelize_reference(&p, &sc, lref.data());
        vcto_voxelize_conservative_attr(&p, &sc, l0.data(), acc.data(), a_alb.data(), a_nrm.data());
        {
            std::vector<uint8_t> slab((size_t)3 * p.V * p.V * 4);
            vcto_voxelize_conservative_zslab(&p, &sc, 2, 5, slab.data());
        }
        std::vector<uint8_t> chain(nchain * 4, 0), aniso(6 * (nchain - nvox) * 4, 0);
        memcpy(chain.data(), l0.data(), nvox * 4);
        vcto_build_mips(chain.data(), V);
        vcto_build_mips_aniso(l0.data(), V, aniso.data());
        const uint64_t bs = vcto_bounce(&p, chain.data(), a_alb.data(), a_nrm.data(), l1.data(), 3);
        const size_t npix = (size_t)w * h;
        std::vector<float> o32(npix * 4), cones(npix * 28);
        std::vector<uint16_t> o16(npix * 4);
        std::vector<uint8_t> steps(npix * 7);
        const uint64_t t1 = vcto_trace(&p, chain.data(), planes.data(), npix, o32.data(), o16.data(), steps.data(), cones.data(), 1);
        const uint64_t t2 = vcto_trace_aniso(&p, chain.data(), aniso.data(), planes.data(), npix, o32.data(), nullptr, nullptr, nullptr, 3);
        p.wrap_repeat = 0;
        const uint64_t t3 = vcto_trace(&p, chain.data(), planes.data(), npix, nullptr, nullptr, nullptr, nullptr, 2);
        p.wrap_repeat = 1;
        printf("scene %d: %d tris, bounce steps %llu, trace steps %llu / %llu / %llu\n", kind, ntri,
               (unsigned long long)bs, (unsigned long long)t1, (unsigned long long)t2, (unsigned long long)t3);
        vcth_scene_destroy(s);
    }
    // OBJ reader incl. error paths
    const char* path = "/tmp/vct_sanitize_cube.obj";
    FILE* f = fopen(path, "w");
    fputs("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 1 1\nvn 0 0 1\nusemtl m\nf 1/1/1 2/2/1 3/3/1 4\nf -4 -3 -2\n", f);
    fclose(f);
    char err[256];
    vcth_scene* o = vcth_scene_load_obj(path, err);
    if (!o || vcth_scene_num_triangles(o) != 3) { printf("obj load failed: %s\n", err); return 1; }
    vcth_scene_destroy(o);
    if (vcth_scene_load_obj("/nonexistent.obj", err) != nullptr) return 1;
    f = fopen(path, "w"); fputs("v 0 0 0\nf 1 2 9\n", f); fclose(f);
    if (vcth_scene_load_obj(path, err) != nullptr) return 1;
    // small helpers
    float out4[4];
    const float pos3[3] = {1.0f, -2.0f, 3.0f};
    std::vector<uint8_t> tiny(vcto_chain_texels(8) * 4, 77);
    p.V = 8;
    vcto_sample(&p, tiny.data(), pos3, 1.5f, out4);
    vcto_sample(&p, tiny.data(), pos3, 99.0f, out4);
    vcto_sample(&p, tiny.data(), pos3, -1.0f, out4);
    { const float uvw[3] = {0.25f, 1.75f, -0.5f}; vcto_texture_lod(&p, tiny.data(), uvw, 0.7f, out4); }
    if (vcto_f32_to_f16(65520.0f) != 0x7c00 || vcto_f16_to_f32(0x3c00) != 1.0f) return 1;
    if (const char* dir = getenv("VCT_SANITIZE_IMAGES")) if (fuzz_images(dir)) return 2;
    printf("sanitize_check ok\n");
    return 0;
}
