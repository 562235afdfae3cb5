"""The C-ABI library must load and export every symbol include/kai_core.h declares (no compute without a GPU)."""
import ctypes as C
import os
import re

import pytest

import kai_testlib as T


def declared_symbols():
    hdr = open(os.path.join(T.ROOT, "include", "kai_core.h")).read()
    return sorted(set(re.findall(r"^(?:int|const char\*)\s+(kai_[a-z_]+)\s*\(", hdr, flags=re.M)))


def test_library_exports_header_symbols():
    lib = T.pkg.load_library()
    syms = declared_symbols()
    assert len(syms) >= 13, syms
    for s in syms:
        assert hasattr(lib, s), f"libkai_core.so does not export {s}"
    assert set(T.pkg.core.EXPORTS) == set(syms)
    assert b"gfx950" in lib.kai_version()


def test_struct_layouts_match_header():
    # sizes computed by hand from include/kai_core.h (LP64)
    assert C.sizeof(T.abi.KaiOp) == 32  # ABI v5: + stmt, pad
    assert C.sizeof(T.abi.KaiQueueShare) == 6 * 3 * 8
    assert C.sizeof(T.abi.KaiNodeState) == 3 * 8 * 8
    assert C.sizeof(T.abi.KaiActionStats) == 6 * 8 + 2 * 8 + 8 * 8
    assert C.sizeof(T.abi.KaiConfig) == 4 + 4 + 4 + 4 + 8 + 8 + 4 * 6 + 8 + 16 + 4 + 28 + 3 * 8 + 4 + 4  # incl. alignment padding after cpu_strategy; minruntime fields last


def test_no_device_fails_loudly():
    """Without a HIP device the library must refuse (KAI_ERR_NO_DEVICE) — there is no CPU fallback."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = T.pkg.load_library()
    cfg = T.abi.default_config()
    h = C.c_void_p()
    ids = (C.c_int * 1)(0)
    rc = lib.kai_core_create(C.byref(cfg), 1, ids, C.byref(h))
    assert rc == -2, rc
    with pytest.raises(T.pkg.KaiError):
        T.pkg.KaiCore(cfg)


def test_ctypes_mirror_matches_the_header_layout():
    """sizeof / offsetof as the C compiler sees include/kai_core.h (probed through the oracle library, which includes the header) against
    the ctypes structures of kai-scheduler_amd/abi.py: a drifted mirror would hand the device library garbage pointers."""
    lib = T.Oracle.lib()
    S, Cf = T.abi.KaiSnapshotSoA, T.abi.KaiConfig
    assert lib.kai_oracle_layout(0) == C.sizeof(S)
    assert lib.kai_oracle_layout(1) == S.node_gpu_memory.offset
    assert lib.kai_oracle_layout(2) == S.job_signature.offset
    assert lib.kai_oracle_layout(3) == S.class_fit.offset
    assert lib.kai_oracle_layout(7) == S.n_groups.offset
    assert lib.kai_oracle_layout(4) == C.sizeof(Cf)
    assert lib.kai_oracle_layout(5) == Cf.now_ns.offset
    assert lib.kai_oracle_layout(6) == C.sizeof(T.abi.KaiActionStats)
