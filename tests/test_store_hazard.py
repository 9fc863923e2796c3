"""CPU test of the build check that guards the 16-byte store-data hazard: the gfx950 assembly
of integrate.hip, as `make` leaves it, passes tools/check_store_hazard.py; the same assembly with the guard taken out,
or with a write to a store's data register moved in front of it, does not.  (The GPU-side regression of the hazard
itself is tests/test_integrate_batch_gpu.py::test_batch_512_equals_single_frame_kernel.)"""
import importlib.util
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASM = os.path.join(ROOT, "yak_amd", "csrc", "integrate.gfx950.s")


def checker():
    spec = importlib.util.spec_from_file_location("check_store_hazard", os.path.join(ROOT, "tools", "check_store_hazard.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_every_16_byte_store_of_the_integrate_kernels_is_guarded():
    assert os.path.exists(ASM), "integrate.gfx950.s missing: run __graft_entry__.build() (make -C yak_amd/csrc)"
    stores, problems = checker().check(open(ASM).read())
    assert stores >= 8, stores  # single-frame and batched kernels, nt and plain variants
    assert problems == []


def test_the_check_fails_without_the_guard():
    text = open(ASM).read()
    c = checker()
    stores, problems = c.check(text.replace("s_nop 7", "s_nop 0"))
    assert len(problems) == stores
    # a VALU write into the store's first data register right behind it, in front of the s_nop
    m = re.search(r"^(\s*)buffer_store_dwordx4\s+v\[(\d+):\d+\].*$", text, flags=re.M)
    bad = text[:m.end()] + "\n%sv_cndmask_b32_e32 v%s, v1, v2, vcc" % (m.group(1), m.group(2)) + text[m.end():]
    _, problems = c.check(bad)
    assert len(problems) == 1 and "writes v[%s]" % m.group(2) in problems[0]
    assert c.check("s_endpgm\n") == (0, [])
