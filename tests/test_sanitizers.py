"""Host code under AddressSanitizer / UndefinedBehaviorSanitizer (CPU build only; GPU sanitizers are not available on the
pool): the sparse-grid generator, the reader / writer of the reference's cereal table file on truncated and corrupted
input, and the sign-orbit decomposition.  Reference formats: quadrature/saveSparseGHWeightMap.h:14-51,
helpers/SerializeEigenMaps.h:195-224."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gaussianvi_amd", "csrc")
SAN = ["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=all"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")


def _build(tmp_path, name, sources):
    exe = str(tmp_path / name)
    r = subprocess.run(SAN + sources + ["-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_table_reader_writer_and_generator_under_sanitizers(tmp_path):
    exe = _build(tmp_path, "table_fuzz", [os.path.join(ROOT, "tests", "stubs", "table_fuzz.cpp"), os.path.join(CSRC, "spgh.cpp"),
                                          os.path.join(CSRC, "table_io.cpp")])
    scratch = tmp_path / "scratch"
    scratch.mkdir()
    r = subprocess.run([exe, str(scratch)], capture_output=True, text=True, env=ENV, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-2000:] + r.stderr[-4000:]


def test_orbit_decomposition_and_walk_under_sanitizers(tmp_path):
    exe = _build(tmp_path, "orbits_check", [os.path.join(ROOT, "tests", "stubs", "orbits_check.cpp"), os.path.join(CSRC, "spgh.cpp")])
    r = subprocess.run([exe, "5", "2", "4", "3", "6", "5", "12", "5", "6", "7"], capture_output=True, text=True, env=ENV, timeout=600)
    assert r.returncode == 0 and r.stdout.count("ok ") == 5, r.stdout[-2000:] + r.stderr[-4000:]
