"""What kai_session_open leaves in device memory, on a machine without a GPU.

kai_session_open does not SEND every array of the session: constants of the snapshot (the shared-GPU model's per-pod arrays when nothing asks for a shared GPU, the identity tables
of a snapshot without sub-group trees, "no nominated node", absent optional arrays) are written on the device by memsets and device-to-device copies.  This test runs the library's
HOST side for real — kai_core.hip compiled host-only (hipcc --cuda-host-only, seconds) and linked with tests/host_sim/fake_hip.cpp instead of libamdhip64: device memory is host
memory, copies and memsets happen at the call, kernels do nothing — and compares the image a default open leaves in "device" memory with the image of KAI_OPEN_FULL_UPLOADS=1
(every array sent from the host, as the library did before) and with the image of KAI_OPEN_NO_STAGING=1 (no pinned staging of the snapshot's own arrays): byte for byte the same on
every shape, with fewer bytes over the bus."""
import json
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

DRIVER = r'''
import ctypes as C, json, sys
sys.path.insert(0, ROOT + "/tests"); sys.path.insert(0, ROOT)
import kai_testlib as T
lib = C.CDLL(LIB)
lib.kai_last_error.restype = C.c_char_p; lib.kai_last_error.argtypes = [C.c_void_p]
syn = T.pkg.synth
def shapes():
    yield "C1", syn.config(0)[:2]
    yield "C2", syn.config(1, 1.0)[:2]
    yield "C3 at 30 %", syn.config(2, 0.3)[:2]
    yield "C5 at 5 %", syn.config(4, 0.05)[:2]
    yield "C5 at 20 %", syn.config(4, 0.2)[:2]   # (its own arrays exceed 4 MB: sent from pinned staging)
    yield "C5-mixed at 3 %", syn.config(4, 0.03, mixed=True)[:2]
    s, c, _ = syn.config(1, 0.5); syn.add_fractions(s, 7, frac=0.3); yield "C2 at 50 % with fractions", (s, c)
    s, c, _ = syn.config(1, 0.2); import numpy as np; s.arrays["pod_gpu_portion"] = np.zeros(s.n_pods); yield "C2 at 20 %, a portion array of zeros", (s, c)
    for seed in (11, 12):
        for ci, (snap, cfg, acts) in enumerate(T.broad_case(seed)):
            yield f"broad {seed}/{ci}", (snap, cfg)
out = {}
for name, (snap, cfg) in shapes():
    h = C.c_void_p()
    assert lib.kai_core_create(C.byref(cfg), 1, None, C.byref(h)) == 0
    st = snap.as_struct()
    rc = lib.kai_session_open(h, C.byref(st))
    assert rc == 0, (name, rc, lib.kai_last_error(h))
    img = (C.c_uint64 * 9)(); lib.fakehip_image(img)
    out[name] = [int(x) for x in img]
    lib.kai_core_destroy(h)
print(json.dumps(out))
'''


@pytest.fixture(scope="module")
def fake_lib(tmp_path_factory):
    if not os.path.exists(HIPCC) or shutil.which("g++") is None:
        pytest.skip("hipcc / g++ not available")
    d = tmp_path_factory.mktemp("fakehip")
    obj, shim, stub, lib = str(d / "kai_core_host.o"), str(d / "fake_hip.o"), str(d / "fatbin_stub.c"), str(d / "libkai_core_fakehip.so")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "--cuda-host-only", "-O1", "-std=c++17", "-ffp-contract=off", "-fPIC", "-c", "-o", obj,
                           os.path.join(ROOT, "kai-scheduler_amd", "csrc", "kai_core.hip")], stderr=subprocess.DEVNULL)
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-I/opt/rocm/include", "-c", "-o", shim, os.path.join(ROOT, "tests", "host_sim", "fake_hip.cpp")])
    undefined = subprocess.check_output(["nm", obj], text=True)
    fat = [ln.split()[-1] for ln in undefined.splitlines() if " U __hip_fatbin" in ln]
    with open(stub, "w") as f:
        f.write("".join(f"const char {name}[16] = {{0}};\n" for name in fat))
    subprocess.check_call(["g++", "-shared", "-o", lib, obj, shim, "-x", "c", stub, "-lpthread", "-ldl"])
    return lib


def _images(lib, env):
    e = dict(os.environ); e.pop("KAI_OPEN_FULL_UPLOADS", None); e.update(env)
    code = f"ROOT = {ROOT!r}\nLIB = {lib!r}\n" + DRIVER
    r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_device_image_of_an_open_does_not_depend_on_what_is_sent(fake_lib):
    lean = _images(fake_lib, {})
    full = _images(fake_lib, {"KAI_OPEN_FULL_UPLOADS": "1"})
    assert lean.keys() == full.keys() and len(lean) >= 9
    saved = 0
    for name in lean:
        h, n_alloc, n_bytes, launches, h2d, h2d_bytes, d2d, memsets, pinned = lean[name]
        fh, fn_alloc, fn_bytes, flaunches, fh2d, fh2d_bytes, fd2d, fmemsets, fpinned = full[name]
        assert (h, n_alloc, n_bytes, launches) == (fh, fn_alloc, fn_bytes, flaunches), name   # the same image, the same allocations, the same kernels
        assert h2d_bytes <= fh2d_bytes, name
        saved += fh2d_bytes - h2d_bytes
    assert saved > 0  # some shape has constants the default open does not send
    # a snapshot without shared GPUs and sub-group trees sends markedly less
    assert lean["C5 at 5 %"][5] < 0.7 * full["C5 at 5 %"][5]
    # the snapshot's own arrays go up from pinned staging when they are large (UploadStage): the same image as sending them from where they lie
    assert lean["C5 at 20 %"][8] > (4 << 20)  # (the staging buffer was allocated: the pinned way ran)
    plain = _images(fake_lib, {"KAI_OPEN_NO_STAGING": "1"})
    assert all(plain[k][:4] == lean[k][:4] for k in lean) and plain["C5 at 20 %"][8] == 0
    # KAI_HOST_POOL=0 / one host thread: the same images (the preparation's outputs do not depend on how its loops run)
    assert _images(fake_lib, {"KAI_HOST_THREADS": "1"}) == lean
    assert _images(fake_lib, {"KAI_HOST_POOL": "0", "KAI_HOST_THREADS": "5"}) == lean


REFUSAL = r'''
import ctypes as C, sys
sys.path.insert(0, ROOT + "/tests"); sys.path.insert(0, ROOT)
import kai_testlib as T
lib = C.CDLL(LIB)
lib.kai_last_error.restype = C.c_char_p; lib.kai_last_error.argtypes = [C.c_void_p]
snap, cfg, _ = T.pkg.synth.config(4, 0.2)
h = C.c_void_p(); assert lib.kai_core_create(C.byref(cfg), 1, None, C.byref(h)) == 0
good = snap.as_struct()
assert lib.kai_session_open(h, C.byref(good)) == 0
bad_job = snap.arrays["pod_job"].copy(); bad_job[-3] = snap.n_jobs + 5
keep = snap.arrays["pod_job"]; snap.arrays["pod_job"] = bad_job
st = snap.as_struct(); rc = lib.kai_session_open(h, C.byref(st)); msg = lib.kai_last_error(h).decode()
assert rc != 0 and "pod_job out of range" in msg, (rc, msg)          # refused by the preparation, after the snapshot's arrays were staged
snap.arrays["pod_job"] = keep
st = snap.as_struct(); st.pod_req = None
rc = lib.kai_session_open(h, C.byref(st)); msg = lib.kai_last_error(h).decode()
assert rc != 0 and "a required pod array is NULL" in msg, (rc, msg)  # refused before anything reads the arrays
assert lib.kai_session_open(h, C.byref(good)) == 0                    # and the handle opens the next session
assert lib.kai_core_destroy(h) == 0
print("refusals ok")
'''


def test_a_refused_snapshot_leaves_the_handle_usable(fake_lib):
    code = f"ROOT = {ROOT!r}\nLIB = {fake_lib!r}\n" + REFUSAL
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "refusals ok" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])


SEQUENCE = r'''
import ctypes as C, sys, os
sys.path.insert(0, ROOT + "/tests"); sys.path.insert(0, ROOT)
import kai_testlib as T
lib = C.CDLL(LIB)
syn = T.pkg.synth
shapes = [("C5 at 10 %", syn.config(4, 0.1)[:2]), ("C1", syn.config(0)[:2]), ("C5 at 20 %", syn.config(4, 0.2)[:2]), ("C2", syn.config(1, 1.0)[:2]), ("C5-mixed at 3 %", syn.config(4, 0.03, mixed=True)[:2])]
s, c, _ = syn.config(1, 0.5); syn.add_fractions(s, 7, frac=0.3); shapes.append(("C2 at 50 % with fractions", (s, c)))
shapes += [(f"broad 12/{ci}", (snap, cfg)) for ci, (snap, cfg, acts) in enumerate(T.broad_case(12))] + [shapes[0], shapes[2]]
one = MODE == "one handle"
h = C.c_void_p()
for i, (name, (snap, cfg)) in enumerate(shapes):
    if not one or i == 0:
        h = C.c_void_p(); assert lib.kai_core_create(C.byref(cfg), 1, None, C.byref(h)) == 0
    st = snap.as_struct()
    assert lib.kai_session_open(h, C.byref(st)) == 0, name
    sys.stderr.write("== %d %s\n" % (i, name)); sys.stderr.flush()
    if not one: assert lib.kai_core_destroy(h) == 0
'''


def test_a_handle_that_opens_session_after_session_fills_the_context_like_a_fresh_one(fake_lib):
    """One handle over a sequence of sessions of very different sizes (its kept preparation objects, its device slabs and its pinned staging buffer all reused and regrown) against a
    fresh handle per session: every array of the session context holds the same bytes after the open (KAI_OPEN_DIGEST).  (The broad-campaign cases share one kai_config shape; the
    handle's configuration is the first session's — as a scheduler's handle keeps its configuration over the cycles.)"""
    import re

    def digests(mode):
        e = dict(os.environ); e["KAI_OPEN_DIGEST"] = "1"
        code = f"ROOT = {ROOT!r}\nLIB = {fake_lib!r}\nMODE = {mode!r}\n" + SEQUENCE
        r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        out, cur = {}, {}
        for ln in r.stderr.splitlines():
            m = re.match(r"kai open digest: field (\d+) bytes (\d+) fnv (\w+)", ln)
            if m: cur[int(m.group(1))] = (int(m.group(2)), m.group(3))
            elif ln.startswith("== "): out[ln[3:]] = cur; cur = {}
        return out
    # arrays the open allocates WITHOUT writing them (the operation log, the output staging, the GPU-group ids a kernel of the open resets): whatever the slab held — not part of the statement
    probe = os.path.join(os.path.dirname(fake_lib), "ctx_offsets")
    with open(probe + ".cpp", "w") as f:
        f.write('#include <cmath>\n#include <math.h>\n#include <cstddef>\n#include <cstdio>\n#define KAI_SHARED_GPUS 1\n#include "%s"\nint main() { std::printf("%%zu %%zu %%zu\\n", offsetof(kai::KaiCtx, ops), offsetof(kai::KaiCtx, out_ops), offsetof(kai::KaiCtx, ng_id)); }\n'
                % os.path.join(ROOT, "kai-scheduler_amd", "csrc", "kai_engine.hpp"))
    subprocess.check_call(["g++", "-std=c++17", "-Wno-invalid-offsetof", "-o", probe, probe + ".cpp"])
    scratch = {int(x) for x in subprocess.check_output([probe], text=True).split()}
    one, fresh = digests("one handle"), digests("fresh handles")
    assert one.keys() == fresh.keys() and len(one) >= 12
    for k in one:
        assert len(one[k]) > 100 and one[k].keys() == fresh[k].keys(), k
        differing = {f for f in one[k] if one[k][f] != fresh[k][f]}
        assert differing <= scratch, (k, differing, scratch)


def test_the_shipped_library_under_the_stand_in_runtime(fake_lib):
    """The library as it ships (kai-scheduler_amd/csrc/libkai_core.so: hipcc -O3, device code embedded, linked against libamdhip64) with its HIP calls bound to the stand-in runtime
    by LD_PRELOAD: the open fills every array of the session context with the same bytes as the host-only build of this test, and what it leaves in device memory does not depend on
    KAI_OPEN_FULL_UPLOADS either.  (The binary that goes to the GPU box, not a rebuild of its source.)"""
    import re
    shipped = os.path.join(ROOT, "kai-scheduler_amd", "csrc", "libkai_core.so")
    if not os.path.exists(shipped):
        pytest.skip("libkai_core.so is not built")
    shim = os.path.join(os.path.dirname(fake_lib), "libfakehip.so")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-I/opt/rocm/include", "-o", shim, os.path.join(ROOT, "tests", "host_sim", "fake_hip.cpp"), "-lpthread"])

    def run(lib, preload, env):
        e = dict(os.environ); e.pop("KAI_OPEN_FULL_UPLOADS", None); e["KAI_OPEN_DIGEST"] = "1"; e.update(env)
        d = DRIVER.replace("out[name] = [int(x) for x in img]", "out[name] = [int(x) for x in img]; sys.stderr.write('== ' + name + chr(10)); sys.stderr.flush()")
        if preload:
            e["LD_PRELOAD"] = shim
            d = d.replace("lib = C.CDLL(LIB)", f"lib = C.CDLL(LIB); shim = C.CDLL({shim!r})").replace("lib.fakehip_image(img)", "shim.fakehip_image(img)")
        r = subprocess.run([sys.executable, "-c", f"ROOT = {ROOT!r}\nLIB = {lib!r}\n" + d], env=e, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        fields, cur = {}, {}
        for ln in r.stderr.splitlines():
            m = re.match(r"kai open digest: field (\d+) bytes (\d+) fnv (\w+)", ln)
            if m: cur[int(m.group(1))] = (int(m.group(2)), m.group(3))
            elif ln.startswith("== "): fields[ln[3:]] = cur; cur = {}
        return fields, json.loads(r.stdout.strip().splitlines()[-1])
    ship_fields, ship_img = run(shipped, True, {})
    host_fields, _ = run(fake_lib, False, {})
    assert ship_fields.keys() == host_fields.keys() and len(ship_fields) >= 9
    for k in ship_fields:
        assert len(ship_fields[k]) > 100 and ship_fields[k] == host_fields[k], k
    _, full_img = run(shipped, True, {"KAI_OPEN_FULL_UPLOADS": "1"})
    assert all(ship_img[k][:4] == full_img[k][:4] for k in ship_img)
    assert sum(full_img[k][5] for k in full_img) > sum(ship_img[k][5] for k in ship_img)
