"""The CPU checkers of the raster input stages (oracle/vct_oracle_raster.cpp) driven with a scene of
voxel-cone-tracing_amd/scene.py.  Test infrastructure: only tests/ may import this."""
import numpy as np

from oracle import pyoracle


# mipmaps: the material textures carry mip chains and are sampled with implicit derivatives (the reference's sampler
# state and the GPU library's default, config.texture_mipmaps = 1); False = level 0 bilinear (the rounds 1-2 fixtures)
def mesh_of(scene, model_scale=0.05, mipmaps=True):
    return pyoracle.make_mesh(scene.pos, scene.material, scene.albedo, scene.specular, scene.frames(), scene.uv,
                              scene.mat_tex, scene.textures, model_scale, mipmaps=mipmaps)


def shadow_map(sc, scene, light_dir, size):
    """Returns (depth [size,size] fp32, light_vp row-major 4x4) -- DrawDepthTexture on the CPU."""
    vp = sc.light_view_proj(light_dir)
    return pyoracle.render_shadow_map(mesh_of(scene), vp, size), vp.reshape(4, 4).T.copy()


def gbuffer(sc, scene, cam, w, h, shadow=None, light_vp_row=None, mipmaps=True):
    """planes [23, w*h] -- the raster + non-cone fragment work of Render() on the CPU."""
    vp = sc.camera_view_proj(cam, w, h)
    lvp = None if light_vp_row is None else np.ascontiguousarray(np.asarray(light_vp_row, np.float32).T).reshape(16)
    return pyoracle.render_gbuffer(mesh_of(scene, mipmaps=mipmaps), vp, w, h, shadow, lvp)


def oracle_scene(scene, shadow_depth=None, light_vp_row=None, mipmaps=True):
    """Input of the oracle voxelizers, with the scene's texture coordinates and diffuse textures."""
    return pyoracle.make_scene(scene.pos, scene.material, scene.albedo, shadow_depth=shadow_depth,
                               light_vp=light_vp_row, uv=scene.uv, mat_tex=scene.mat_tex, textures=scene.textures,
                               mipmaps=mipmaps)
