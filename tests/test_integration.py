"""CPU: the drop-in boundary compiled the way a `-DBACKEND=hip` Nova would compile it.
integration/CMakeLists.txt builds integration/boundary_check.cpp against Nova's REAL src/kernel_types and
src/backend/common headers, with the three dispatch headers patched by integration/patch_nova.py (an include
overlay in the build directory; the Nova tree is not touched, nothing of it is copied into this repository) and
integration/hip_backend.cmake for the define, include paths and link.  The translation unit static_asserts sizeof
and offsetof of every shared struct against include/nova_hip.h and instantiates every Accelerator template with
Raytracer's member types and argument lists.  Needs the Nova tree: runs in the build container only."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NOVA = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(os.path.join(NOVA, "src", "kernel_types")), reason="needs the Nova source tree")
@pytest.mark.skipif(shutil.which("cmake") is None, reason="needs cmake")
def test_hip_backend_compiles_against_novas_own_headers(tmp_path):
    assert os.path.exists(os.path.join(ROOT, "nova_amd", "libnova_hip.so")), "run __graft_entry__.build() first"
    build = str(tmp_path / "build")
    cfg = subprocess.run(["cmake", "-S", os.path.join(ROOT, "integration"), "-B", build, "-DNOVA_ROOT=" + NOVA,
                          "-DNOVA_HIP_ROOT=" + ROOT], capture_output=True, text=True, timeout=300)
    assert cfg.returncode == 0, cfg.stdout[-2000:] + cfg.stderr[-2000:]
    res = subprocess.run(["cmake", "--build", build], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    for rel in ("src/backend/accelerator.hpp", "src/backend/types.hpp", "src/backend/common/utils/utils.hpp"):
        text = open(os.path.join(build, "overlay", rel)).read()
        assert "#elif defined(BACKEND_HIP)" in text and "backend/hip/" in text
    assert os.path.exists(os.path.join(build, "boundary_check"))


def test_patch_script_on_a_header_of_that_shape(tmp_path):
    """patch_nova.py needs no Nova to be tested: a dispatch header of the same shape gets one more branch, once."""
    sys.path.insert(0, os.path.join(ROOT, "integration"))
    import patch_nova
    text = ("#ifdef BACKEND_OPENCL\n  #include \"a/opencl.hpp\"\n#elif defined(BACKEND_CUDA)\n  #include \"a/common.hpp\"\n"
            "  #include \"a/cuda.hpp\"\n#endif\n")
    out = patch_nova.patched(text, ['#include "a/hip.hpp"'])
    assert out == text.replace("#endif\n", "#elif defined(BACKEND_HIP)\n  #include \"a/hip.hpp\"\n#endif\n")
    assert patch_nova.patched(out, ['#include "a/hip.hpp"']) == out
    with pytest.raises(SystemExit):
        patch_nova.patched("no dispatch block here\n", [])
