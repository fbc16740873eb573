"""The host-side C and C++ of the product (asset preparation, prepared-mesh files, the scene flattener) under
AddressSanitizer and UndefinedBehaviorSanitizer: two small programs in tests/c_host are compiled together with the sources
they drive and run on the CPU.  They feed well-formed input, truncated and randomly damaged prepared-mesh files, and scenes
whose indices point outside their tables; a sanitizer report, a leak or a failed check fails the test."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "raytracing_amd", "csrc")
HERE = os.path.join(ROOT, "tests", "c_host")
SAN = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
       "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]


def run(cmd, **kw):
    done = subprocess.run(cmd, capture_output=True, text=True, timeout=600, **kw)
    if done.returncode != 0 and "LeakSanitizer has encountered a fatal error" in done.stderr and "env" in kw:
        # the leak checker cannot stop the process's threads under some supervisors (it needs ptrace): the other checks still run
        env = dict(kw["env"])
        env["ASAN_OPTIONS"] = env["ASAN_OPTIONS"].replace("detect_leaks=1", "detect_leaks=0")
        done = subprocess.run(cmd, capture_output=True, text=True, timeout=600, **{**kw, "env": env})
    if done.returncode != 0 and "ASan runtime does not come first" in done.stderr:
        pytest.skip("another preloaded library keeps the sanitizer runtime from starting in this environment")
    assert done.returncode == 0, (cmd[0], done.stdout[-2000:], done.stderr[-4000:])
    return done.stdout


@pytest.fixture(scope="module")
def build_dir(tmp_path_factory):
    if not shutil.which("gcc") or not shutil.which("g++"):
        pytest.skip("no host compiler")
    return tmp_path_factory.mktemp("sanitized")


def sanitized_env():
    env = dict(os.environ)
    env["ASAN_OPTIONS"] = "detect_leaks=1:abort_on_error=0"
    env["UBSAN_OPTIONS"] = "print_stacktrace=1"
    return env


def test_host_asset_code_under_sanitizers(build_dir):
    exe = str(build_dir / "host_assets_check")
    run(["gcc", "-std=c11", *SAN, os.path.join(HERE, "host_assets_check.c"), os.path.join(CSRC, "rt_host_assets.c"),
         "-o", exe, "-lm", "-lpthread"])
    assert run([exe, str(build_dir / "check.mesh")], env=sanitized_env()).strip() == "ok"


def test_scene_flattener_under_sanitizers(build_dir):
    obj = str(build_dir / "host_assets.o")
    exe = str(build_dir / "flatten_check")
    run(["gcc", "-std=c11", *SAN, "-c", os.path.join(CSRC, "rt_host_assets.c"), "-o", obj])
    run(["g++", "-std=c++17", *SAN, os.path.join(HERE, "flatten_check.cpp"), os.path.join(CSRC, "rt_flatten.cpp"), obj,
         "-o", exe, "-lm", "-lpthread"])
    assert run([exe], env=sanitized_env()).strip() == "ok"
