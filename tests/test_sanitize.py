"""The CPU-side code (oracle + host library: rasterisers, scene builders, OBJ reader) once under
AddressSanitizer + UBSan (GPU sanitizers are not available on the pool)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_and_host_library_are_clean_under_asan_ubsan():
    out = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "sanitize"], capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "sanitize_check ok" in out.stdout
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr
