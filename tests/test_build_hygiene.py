"""Static checks of the shipped kernel build (no GPU): compiler-configuration hazards found in rounds 1-2.

`-mllvm -disable-machine-cse` (tried as a tuning flag in round 1) produced a kernel that flagged every locus with
VLR_LOCUS_UNDERFLOW.  Root cause (round 2, tools/repro/nocse_exp.hip): without MachineCSE this compiler materialises
64-bit floating-point constants as `s_mov_b64 sN, <64-bit literal>`; gfx950 cannot encode 64-bit literals, the object
keeps the low 32 bits (+inf becomes 0, the range thresholds of exp() become denormals).  The textual ISA shows the full
literal, so the pattern can be searched for: the shipped flags must not produce it."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"
BAD = re.compile(r"^\s*s_\w+_b64\s+[^;\n]*\b0x[0-9a-fA-F]{9,16}\b", re.M)

pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")


def _asm(src, flags, tmp_path, name):
    out = str(tmp_path / name)
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "--cuda-device-only", "-S", src, "-o", out] + flags, stderr=subprocess.DEVNULL)
    with open(out) as fh:
        return fh.read()


def _makefile_flags():
    with open(os.path.join(ROOT, "varlociraptor_amd", "csrc", "Makefile")) as fh:
        for line in fh:
            if line.startswith("CXXFLAGS"):
                return [f for f in line.split("=", 1)[1].split() if f not in ("-fPIC", "-Wall")]
    raise AssertionError("CXXFLAGS not found")


def test_shipped_flags_do_not_use_disable_machine_cse():
    assert "-disable-machine-cse" not in _makefile_flags()


def test_shipped_isa_has_no_truncated_64bit_scalar_literals(tmp_path):
    text = _asm(os.path.join(ROOT, "varlociraptor_amd", "csrc", "vlr_kernels.hip"), _makefile_flags(), tmp_path, "vlr.s")
    assert "vlr_call_kernel" in text
    hits = BAD.findall(text)
    assert not hits, hits[:5]


def test_disable_machine_cse_reproducer_shows_the_compiler_defect(tmp_path):
    src = os.path.join(ROOT, "tools", "repro", "nocse_exp.hip")
    good = _asm(src, ["-O3"], tmp_path, "ok.s")
    bad = _asm(src, ["-O3", "-mllvm", "-disable-machine-cse"], tmp_path, "nocse.s")
    assert not BAD.findall(good)
    if not BAD.findall(bad):
        pytest.skip("this compiler no longer emits 64-bit scalar literals without MachineCSE")
    # the compiler's own assembler refuses what its code generator printed
    clang = "/opt/rocm/lib/llvm/bin/clang"
    if os.path.exists(clang):
        r = subprocess.run([clang, "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", str(tmp_path / "nocse.s"), "-o", str(tmp_path / "nocse.o")],
                           capture_output=True, text=True)
        assert r.returncode != 0 and "invalid operand" in r.stderr


def test_built_libraries_carry_the_id_of_the_current_sources():
    """libvlr.so and the build-matrix variants must come from the sources in the tree (vlr_build_id): a stale variant would
    make tests/test_gpu_build_matrix.py compare two different kernels."""
    import ctypes
    from varlociraptor_amd import engine
    engine.build()
    engine.build_matrix()
    want = engine.source_id()
    paths = [engine.LIB_PATH] + [os.path.join(engine.MATRIX_DIR, "libvlr_%s.so" % n) for n in engine.MATRIX_LIBS]
    for path in paths:
        L = ctypes.CDLL(path)
        L.vlr_build_id.restype = ctypes.c_char_p
        assert L.vlr_build_id().decode() == want, path


def test_no_binary_artefacts_are_tracked():
    """Built code objects, offload-bundler extracts and libraries stay out of the history."""
    if not os.path.isdir(os.path.join(ROOT, ".git")) or shutil.which("git") is None:
        pytest.skip("not a git checkout")
    files = subprocess.run(["git", "-C", ROOT, "ls-files", "varlociraptor_amd", "oracle", "include", "tools"], capture_output=True, text=True).stdout.split()
    bad = [f for f in files if re.search(r"\.(so|o|a|hsaco|co)(\.|$)|hipv4-|host-x86_64", f)]
    assert not bad, bad


def _asm_statements(text):
    """(template, output constraints) of every asm statement of a source text (templates are adjacent string literals)."""
    out = []
    for m in re.finditer(r"\basm\s*(?:volatile)?\s*\(", text):
        i = m.end()
        depth, j = 1, i
        while depth and j < len(text):
            depth += {"(": 1, ")": -1}.get(text[j], 0)
            j += 1
        body = text[i:j - 1]
        parts = re.split(r"(?<!:):(?!:)", body)   # template : outputs : inputs : clobbers
        template = "".join(re.findall(r'"((?:[^"\\]|\\.)*)"', parts[0]))
        outputs = re.findall(r'"([^"]*)"\s*\(', parts[1]) if len(parts) > 1 else []
        out.append((template, outputs))
    return out


def test_asm_statements_of_the_call_kernel_carry_nothing_the_compiler_cannot_check():
    """Round 6: both build-matrix deviations of rounds 4-5 were asm statements of csrc/vlr_kernels.hip.  park_sd read lanes with
    `v_readfirstlane_b32` inside the template — gfx950 needs a wait state between the VALU write of a VGPR and a lane read of it, and
    the hazard recogniser does not look into templates; fresh_sd wrote its first output before reading its last input without
    early-clobber outputs.  Rules for every asm statement of the file: no lane read, DPP or LDS crossbar instruction in a template,
    and a template of more than one instruction declares every output early-clobber."""
    text = open(os.path.join(ROOT, "varlociraptor_amd", "csrc", "vlr_kernels.hip")).read()
    stmts = _asm_statements(text)
    assert len(stmts) >= 6
    for template, outputs in stmts:
        insts = [t.strip() for t in template.replace("\\n", "\n").replace("\\t", " ").split("\n") if t.strip()]
        for ins in insts:
            assert not re.match(r"(v_readlane|v_readfirstlane|v_writelane|ds_swizzle|ds_bpermute|ds_permute|v_permlane)", ins), ins
            assert "dpp" not in ins and "quad_perm" not in ins and "row_" not in ins, ins
        if len(insts) > 1:
            # (one exception by construction: a single output that only the LAST instruction writes — bitonic_pick)
            last_dst = insts[-1].split()[1].rstrip(",") if len(insts[-1].split()) > 1 else ""
            written_early = [o for k, o in enumerate(outputs) if not (len(outputs) == 1 and last_dst == "%0"
                                                                       and not any(re.search(r"\s%0\b,?", " " + i_.split(None, 1)[1].split(",")[0]) for i_ in insts[:-1] if len(i_.split(None, 1)) > 1))]
            for o in written_early:
                assert o.startswith("=&") or o.startswith("+"), (template, outputs)
