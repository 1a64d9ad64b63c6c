"""The C-ABI library: loads without a GPU, exports every symbol declared in include/*.h,
struct layouts match the reference, the numerically sensitive kernels hold no fused
multiply-adds, and nothing silently falls back to the CPU."""
import ctypes as C
import glob
import os
import re
import shutil
import subprocess

import pytest

import sift_amd
from sift_amd import abi, api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    names = set()
    for path in glob.glob(os.path.join(ROOT, "include", "*.h")):
        text = open(path).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        for m in re.finditer(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{}]*\)\s*;", text):
            names.add(m.group(1))
    return names


def test_library_exports_every_declared_symbol():
    L = sift_amd.lib()
    names = declared_functions()
    assert "sift_anatomy" in names and "sift_amd_detect_device" in names and len(names) > 40
    missing = [n for n in sorted(names) if not hasattr(L, n)]
    assert not missing, f"declared in include/ but not exported: {missing}"


def test_struct_layouts():
    assert C.sizeof(abi.SiftParameters) == 56
    assert C.sizeof(abi.Keypoint) == 72
    assert abi.Keypoint.descr.offset == 48 and abi.Keypoint.orihist.offset == 64
    assert C.sizeof(abi.SiftKeypoints) == 16
    assert C.sizeof(abi.Octa) == 32 and C.sizeof(abi.SiftScalespace) == 808
    assert C.sizeof(abi.SiftKeypointStd) == 144
    assert api.KEY_DTYPE.itemsize == 40


def test_default_parameters_and_containers_without_gpu():
    L = sift_amd.lib()
    p = L.sift_assign_default_parameters()
    assert bytes(p.contents) == bytes(abi.SiftParameters.default())
    C.CDLL(None).free(C.cast(p, C.c_void_p))
    lst = L.sift_malloc_keypoints()
    assert lst.contents.size == 0 and lst.contents.capacity == 100
    for k in range(250):
        key = L.sift_malloc_keypoint(8, 4, 36)
        key.contents.x = k
        L.sift_add_keypoint_to_list(key, lst)
    assert lst.contents.size == 250 and lst.contents.capacity == 400
    assert lst.contents.list[249].contents.x == 249.0
    L.sift_free_keypoints(lst)
    ss = L.sift_malloc_scalespace_lowe(3, 3, 64, 48, C.c_float(0.5), C.c_float(0.8))
    oc = ss.contents.octaves[1].contents
    assert (oc.w, oc.h, oc.nSca) == (64, 48, 6) and oc.delta == 1.0
    dog = L.sift_malloc_scalespace_dog_from_scalespace(ss)
    assert dog.contents.octaves[0].contents.nSca == 5
    L.sift_free_scalespace(dog)
    L.sift_free_scalespace(ss)


def test_no_silent_cpu_fallback():
    """Without a GPU the product path must fail loudly, not compute on the host."""
    if sift_amd.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(sift_amd.SiftAmdError):
        sift_amd.Context(64, 48)


def test_product_does_not_reference_the_oracle():
    for path in glob.glob(os.path.join(ROOT, "sift_amd", "**", "*"), recursive=True):
        if os.path.isfile(path) and path.endswith((".py", ".hip", ".h", ".cpp")):
            text = open(path).read()
            assert not re.search(r"^\s*(from|import)\s+oracle|oracle/|liboracle|pyoracle|refrun", text, re.M), \
                f"{path} refers to the test oracle"


def _disassembly(tmp_path):
    """Disassembles the gfx950 code objects embedded in the shipped library."""
    objdump = shutil.which("llvm-objdump") or "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not available")
    so = str(tmp_path / "lib.so")
    shutil.copy(api.LIB_PATH, so)
    subprocess.check_call([objdump, "--offloading", so], stdout=subprocess.DEVNULL)
    cos = glob.glob(so + ".*gfx950*")
    assert cos, "no gfx950 code object in the library"
    return "".join(subprocess.check_output([objdump, "-d", co]).decode() for co in cos)


def test_wide_buffer_stores_keep_their_data_registers(tmp_path):
    """A 12/16-byte buffer store with a scalar offset register must not be followed directly by a
    vector instruction that overwrites its data registers: on gfx950 part of the lanes then store
    the new value, and the compiler does not separate the two by itself (buf_st4 in
    sift_kernels.hip supplies the wait states)."""
    asm = _disassembly(tmp_path)
    current, prev_store, hits, n_stores = None, None, [], 0
    for line in asm.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            current, prev_store = m.group(1), None
            continue
        ins = line.strip().split("//")[0].strip()
        if not ins or not re.match(r"^[a-z]", ins):
            continue
        if prev_store is not None:
            d = re.match(r"^v_(?!cmp|readlane|readfirstlane)\w+\s+(?:v\[(\d+):(\d+)\]|v(\d+))\b", ins)
            if d:
                lo, hi = (int(d.group(3)),) * 2 if d.group(3) else (int(d.group(1)), int(d.group(2)))
                if lo <= prev_store[1] and hi >= prev_store[0]:
                    hits.append((current, prev_store[2], ins))
            prev_store = None
        st = re.match(r"^buffer_store_dwordx[34]\s+v\[(\d+):(\d+)\],.*\bs\d+\s+offen", ins)
        if st:
            n_stores += 1
            prev_store = (int(st.group(1)), int(st.group(2)), ins)
    # the other hazard the compiler leaves to hand-written code here: v_writelane_b32 reading an
    # SGPR in the cycle after the vector compare that wrote it (masks_to_lanes* in sift_kernels.hip,
    # probe: tools/membench/writelane_test.hip)
    prev, n_wl = "", 0
    for line in asm.splitlines():
        ins = line.strip().split("//")[0].strip()
        if not ins or not re.match(r"^[a-z]", ins):
            continue
        if ins.startswith("v_writelane_b32"):
            n_wl += 1
            src = re.match(r"^v_writelane_b32 v\d+, (s(\d+)|vcc_lo|vcc_hi)", ins)
            dst = re.match(r"^v_cmp\w+ (s\[(\d+):(\d+)\]|vcc)", prev)
            if src and dst:
                same = (dst.group(1) == "vcc") == src.group(1).startswith("vcc") and \
                       (src.group(1).startswith("vcc") or int(dst.group(2)) <= int(src.group(2)) <= int(dst.group(3)))
                assert not same, f"v_writelane_b32 reads what the compare directly before it wrote: {prev} / {ins}"
        prev = ins
    assert n_wl >= 36, "the ballot-to-lane writes of the descriptor kernel are not in the disassembly"
    assert n_stores > 100, "no wide buffer stores found: the scan does not match the disassembly"
    assert not hits, f"{len(hits)} wide buffer stores followed by an overwrite of their data, e.g. {hits[:3]}"


def test_blur_kernels_have_no_fused_multiply_add(tmp_path):
    asm = _disassembly(tmp_path)
    # every kernel whose float arithmetic decides discrete extrema or refined positions
    families = ["k_blur_fused", "k_blur_pair", "k_blur_v_any", "k_blur_h_any", "k_oversample", "k_subsample2", "k_sub",
                "k_gradient", "k_extrema", "k_refine"]
    pat = re.compile("|".join(families))
    current, fused, divs, seen = None, {}, {}, set()
    for line in asm.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            current = m.group(1)
            continue
        if current and pat.search(current):
            seen.add(current)
            if re.search(r"\bv_div_fmas_f32", line):
                divs[current] = divs.get(current, 0) + 1
            elif re.search(r"\bv_fmamk_f32 .*0x[4c]f800000", line):
                pass      # 2^32 scaling inside the 64-bit integer division helper, no image data
            elif re.search(r"\bv_(fma|fmac|mad|mac|pk_fma|fmaak|fmamk|madak|madmk)_f32", line):
                fused[current] = fused.get(current, 0) + 1
    # a correctly rounded float division (__fdiv_rn, as the reference's `/`) expands to
    # v_div_scale/v_rcp, five fused steps, v_div_fmas, v_div_fixup: those five are the only
    # fused multiply-adds allowed
    bad = {k: (n, divs.get(k, 0)) for k, n in fused.items() if n != 5 * divs.get(k, 0)}
    for fam in families:
        assert any(re.search(fam + r"(?![a-z_0-9]*_any)" if fam == "k_sub" else fam, k) for k in seen), \
            f"{fam} not found in the disassembly"
    n_fused = sum("k_blur_fused" in k for k in seen)
    assert n_fused >= 64, f"only {n_fused} k_blur_fused instantiations in the disassembly"
    assert not bad, f"fused multiply-adds in bit-exact kernels: {bad}"


def test_quotient_by_a_precomputed_reciprocal_is_the_correctly_rounded_quotient(tmp_path):
    """div_by_uniform (sift_kernels.hip) replaces a / b by q = a*y, q + (a - b*q)*y with y = RN(1/b)
    in the orientation and descriptor kernels.  tests/c_host/div_by_uniform_check.c runs the same
    three operations on the CPU (fmaf / fma are exact FMAs) against '/' on adversarial mantissas, every
    float mantissa for the denominators in use, and 8e7 random operand pairs."""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    exe = str(tmp_path / "dbu")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-o", exe,
                           os.path.join(ROOT, "tests", "c_host", "div_by_uniform_check.c"), "-lm"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout
    assert "float: 0 of" in out.stdout and "double: 0 of" in out.stdout, out.stdout
