"""The C-ABI shared library: it loads, exports everything include/rt_hip.h declares, and its
argument checking and row arithmetic work without a GPU (no compute calls here)."""
import ctypes as C
import pathlib
import re

import pytest

import oracle_lib as O
from raytracer_amd import binding as B
from raytracer_amd import host_scene

ROOT = pathlib.Path(__file__).resolve().parent.parent


def declared_functions(header: pathlib.Path, prefix: str):
    text = re.sub(r"/\*.*?\*/", "", header.read_text(), flags=re.S)
    return sorted(set(re.findall(rf"\b({prefix}\w+)\s*\(", text)))


def test_hip_library_exports_every_declared_symbol():
    lib = B.load_library()
    names = declared_functions(ROOT / "include" / "rt_hip.h", "rt_")
    assert set(names) == set(B.EXPORTS), "binding.EXPORTS and include/rt_hip.h disagree"
    for name in names:
        assert hasattr(lib, name), f"librt_hip.so does not export {name}"
    assert lib.rt_abi_version() == 1


def test_host_library_exports_every_declared_symbol():
    lib = host_scene.load_library()
    names = declared_functions(ROOT / "raytracer_amd" / "host" / "rt_host_capi.h", "rth_")
    assert set(names) == set(host_scene.EXPORTS)
    for name in names:
        assert hasattr(lib, name)


def test_struct_sizes_match_the_header():
    p = B.default_params(640, 480, 16)
    assert p.struct_size == C.sizeof(B.RenderParams) == 19 * 4
    assert C.sizeof(B.World) == 9 * 8          # four (count, pointer) pairs with padding + camera pointer
    assert C.sizeof(B.DeviceInfo) == 128 + 32 + 4 + 4 + 4 + 4 + 8 + 8


def test_default_params_are_the_reference_constants():
    p = B.default_params(1280, 720, 512)
    assert (p.width, p.height, p.samples_per_pixel) == (1280, 720, 512)     # main.cpp:410,415
    assert (p.row_begin, p.row_end) == (0, 720)
    assert p.max_bounces == 8                                                # main.cpp:275
    assert p.min_hit_distance == pytest.approx(0.001) and p.plane_tolerance == pytest.approx(0.001)
    assert p.seed_constant == 262346                                         # main.cpp:370
    assert p.rows_per_order == 1                                             # main.cpp:433
    assert p.intersect_mode == B.INTERSECT_WIDE8 and p.rng_mode == B.RNG_PIXEL


@pytest.mark.parametrize("rows,band_rows,count", [(27, 4, 3), (1080, 8, 8), (720, 8, 5), (9, 8, 4), (16, 8, 2)])
def test_row_band_arithmetic_matches_the_oracle(rows, band_rows, count):
    seen = []
    for index in range(count):
        p = B.default_params(32, rows, 1, band_rows=band_rows, band_index=index, band_count=count)
        op = O.default_params(32, rows, 1, band_rows=band_rows, band_index=index, band_count=count)
        n = B.local_row_count(p)
        assert n == O.lib().rto_local_row_count(C.byref(op))
        mine = list(B.global_rows(p))
        assert mine == [O.lib().rto_global_row(C.byref(op), i) for i in range(n)]
        assert mine == sorted(mine)
        seen += mine
    assert sorted(seen) == list(range(rows))


def test_status_strings():
    lib = B.load_library()
    assert lib.rt_status_string(0) == b"ok"
    assert all(lib.rt_status_string(i) for i in range(7))


def test_create_without_a_gpu_fails_loudly():
    """No device: rt_create reports RT_ERR_NO_DEVICE; nothing falls back to the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(B.RtError) as err:
        B.Context(0)
    assert err.value.status == 2


def test_header_is_plain_c_and_a_c_program_links(tmp_path):
    """include/rt_hip.h is the boundary for any language with a C FFI: it compiles as C89-with-stdint and as C++,
    and a C program linked against librt_hip.so can call the entry points that need no GPU."""
    import subprocess
    src = tmp_path / "use.c"
    src.write_text(r'''
#include <stdio.h>
#include <string.h>
#include "rt_hip.h"
int main(void) {
    rt_render_params p;
    rt_default_params(&p, 1280, 720, 512);
    if (p.struct_size != sizeof p || p.max_bounces != 8 || p.seed_constant != 262346u) return 1;
    p.band_rows = 8; p.band_index = 1; p.band_count = 4;
    if (rt_local_row_count(&p) != 184 || rt_global_row(&p, 8) != 40) return 2;   /* bands 1, 5, 9, ... of 90 */
    if (rt_abi_version() != 1) return 3;
    if (strlen(rt_status_string(RT_ERR_TOO_LARGE)) == 0) return 4;
    if (rt_render(NULL, &p, NULL, NULL, NULL) == RT_OK) return 5;                /* no context: an error, not a crash */
    printf("ok\n");
    return 0;
}
''')
    include, libdir = ROOT / "include", ROOT / "raytracer_amd"
    exe = tmp_path / "use"
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", f"-I{include}", str(src), "-o", str(exe),
                    f"-L{libdir}", "-lrt_hip", f"-Wl,-rpath,{libdir}"], check=True)
    assert subprocess.run([str(exe)], stdout=subprocess.PIPE, text=True, check=True).stdout.strip() == "ok"
    cpp = tmp_path / "use.cpp"
    cpp.write_text('#include "rt_hip.h"\nint main() { rt_render_params p; rt_default_params(&p, 4, 4, 1); return p.width == 4 ? 0 : 1; }\n')
    subprocess.run(["g++", "-std=c++11", "-Wall", "-Werror", f"-I{include}", str(cpp), "-o", str(tmp_path / "usecpp"),
                    f"-L{libdir}", "-lrt_hip", f"-Wl,-rpath,{libdir}"], check=True)
    assert subprocess.run([str(tmp_path / "usecpp")]).returncode == 0
