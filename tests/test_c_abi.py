"""The C ABI driven from a plain C program (tests/c_abi_smoke.c) with the argument layout of the Julia shim in INTEGRATION.md:
`hs_factor_z` / `hs_ldiv_z` on ComplexF64 data behind typed pointers, the real entry points, the error path, `hs_free`."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(n):
    import __graft_entry__ as G
    import hsamd

    hs = hsamd.load()
    exe = G.build_c_abi_smoke()
    return subprocess.run([exe, hs._lib.LIB_PATH, str(n)], capture_output=True, text=True, timeout=300)


def test_c_program_loads_the_library_and_fails_loudly_without_a_device():
    """dlopen + every symbol the shim binds; on a box without a GPU the library reports 'no device' (exit 2), never a CPU result."""
    from conftest import have_gpu

    r = _run(12)
    if have_gpu():
        assert r.returncode == 0, r.stdout + r.stderr
    else:
        assert r.returncode == 2 and "no device" in r.stderr, r.stdout + r.stderr


@pytest.mark.gpu
def test_c_abi_smoke_complex_and_real():
    r = _run(48)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "C_ABI_SMOKE OK" in r.stdout, r.stdout
