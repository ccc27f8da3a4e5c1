"""The CPU-side code (oracle + host library: rasterisers, scene builders, OBJ reader) once under
AddressSanitizer + UBSan (GPU sanitizers are not available on the pool)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sample_images(d):
    """PNG (several colour types), JPEG (4:4:4, 4:2:0, grey, restart markers), BMP, TGA raw / RLE for the decoder fuzz."""
    try:
        from PIL import Image
    except ImportError:
        return False
    import numpy as np
    r = np.random.default_rng(5)
    img = r.integers(0, 256, (23, 31, 4), dtype=np.uint8)
    img[..., 0] = np.arange(31)[None, :] * 8
    Image.fromarray(img, "RGBA").save(os.path.join(d, "a.png"))
    Image.fromarray(img[..., :3], "RGB").save(os.path.join(d, "b.png"))
    Image.fromarray(img[..., :3], "RGB").quantize(9).save(os.path.join(d, "c.png"))
    Image.fromarray(img[..., 1], "L").save(os.path.join(d, "d.png"))
    Image.fromarray(img[..., :3], "RGB").save(os.path.join(d, "e.jpg"), quality=90, subsampling=0)
    Image.fromarray(img[..., :3], "RGB").save(os.path.join(d, "f.jpg"), quality=60, subsampling=2, restart_marker_rows=1)
    Image.fromarray(img[..., 1], "L").save(os.path.join(d, "g.jpg"), quality=80)
    Image.fromarray(img[..., :3], "RGB").save(os.path.join(d, "h.bmp"))
    Image.fromarray(img, "RGBA").save(os.path.join(d, "i.tga"))
    Image.fromarray(img, "RGBA").save(os.path.join(d, "j.tga"), compression="tga_rle")
    return True


def test_oracle_and_host_library_are_clean_under_asan_ubsan(tmp_path):
    env = dict(os.environ)
    if _sample_images(str(tmp_path)):
        env["VCT_SANITIZE_IMAGES"] = str(tmp_path)      # + 15,000 corrupted copies through the image decoders
    out = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "sanitize"], capture_output=True,
                         text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "sanitize_check ok" in out.stdout
    if "VCT_SANITIZE_IMAGES" in env:
        assert "image decoders: 10 files (10 decoded), 15000 corrupted variants" in out.stdout, out.stdout[-500:]
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr
