"""The boundary is a C ABI: `tests/c_abi/flux_handle_demo.c` is a plain-C99 host program (gcc, no C++, no Python, no torch)
that drives the whole path through the handle API of include/vcloze_hip.h.

* not gpu: it compiles as C and links against libvcloze_hip.so (so the header is C-clean and every symbol it uses exists);
* gpu: it runs on the MI355X, and its output file - one Flux.forward and a 4-step Euler trajectory on procedural weights -
  equals, bit for bit, the same calls made from Python through ctypes over the same weights."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, "tests", "c_abi", "flux_handle_demo.c")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")

IN_CH, OUT_CH, VEC, CTX, D, HEADS, DEPTH, SINGLE, MLP, T, N, STEPS = 384, 64, 64, 128, 256, 2, 2, 2, 1024, 16, 24, 4


def build_demo(out_dir) -> str:
    from visualcloze_amd import hip
    if not os.path.exists(hip.LIB_PATH):
        hip.build()
    exe = os.path.join(str(out_dir), "flux_handle_demo")
    libdir = os.path.dirname(hip.LIB_PATH)
    cmd = ["gcc", "-std=gnu99", "-Wall", "-Werror", "-I" + os.path.join(REPO, "include"), "-isystem", os.path.join(ROCM, "include"), SRC,
           "-o", exe, "-L" + libdir, "-lvcloze_hip", "-L" + os.path.join(ROCM, "lib"), "-lamdhip64",
           "-Wl,-rpath," + libdir, "-Wl,-rpath," + os.path.join(ROCM, "lib")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    return exe


@pytest.mark.skipif(shutil.which("gcc") is None, reason="no gcc")
def test_c_host_program_compiles_and_links_as_c99(tmp_path):
    exe = build_demo(tmp_path)
    assert os.path.getsize(exe) > 0
    # the header alone is valid C (no C++-isms outside the extern "C" guard), strictly
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-x", "c",
                        os.path.join(REPO, "include", "vcloze_hip.h")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


# ---------------------------------------------------------------- the same procedural tensors as the C program
def _fnv(name: str) -> int:
    h = 2166136261
    for ch in name.encode():
        h = ((h ^ ch) * 16777619) & 0xffffffff
    return h


def _bf16_bits(f: np.ndarray) -> np.ndarray:
    u = f.astype(np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7fff + ((u >> 16) & 1)) & 0xffffffff
    return (u >> 16).astype(np.uint16)


def fill(name: str, count: int, scale: float, offset: float = 0.0, dev="cuda:0") -> torch.Tensor:
    k = np.arange(count, dtype=np.uint64)
    w = (np.uint64(_fnv(name)) * np.uint64(1664525) + k * np.uint64(1013904223) + np.uint64(12345)) & np.uint64(0xffffffff)
    f = ((w >> np.uint64(16)).astype(np.float32) - np.float32(32768.0)) * np.float32(1.0 / 32768.0)
    bits = _bf16_bits(f * np.float32(scale))
    if offset != 0.0:
        back = (bits.astype(np.uint32) << 16).view(np.float32)
        bits = _bf16_bits(back + np.float32(offset))
    return torch.from_numpy(bits.view(np.int16).copy()).view(torch.bfloat16).to(dev)


@pytest.mark.gpu
def test_c_host_program_equals_python_calls_bitwise(tmp_path):
    from visualcloze_amd import hip
    L = hip.lib()
    exe = build_demo(tmp_path)
    out_file = tmp_path / "out.bin"
    r = subprocess.run([exe, str(out_file)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    got = torch.from_numpy(np.fromfile(out_file, dtype=np.int16)).view(torch.bfloat16).reshape(2 + STEPS, N, OUT_CH)

    keep = []
    h = C.c_void_p()
    cfg = hip.FluxConfig(IN_CH, OUT_CH, VEC, CTX, D, HEADS, DEPTH, SINGLE, MLP, 1, (C.c_int32 * 3)(16, 56, 56), 10000)
    hip._check(L.vc_flux_create(C.byref(cfg), C.byref(h)), "create")

    def bind_linear(name, rows, cols):
        w, b = fill(name + ".weight", rows * cols, 0.06).reshape(rows, cols), fill(name + ".bias", rows, 0.05)
        keep.extend((w, b))
        hip._check(L.vc_flux_bind_weight(h, name.encode(), w.data_ptr(), b.data_ptr(), rows, cols, cols), name)

    def bind_scale(name):
        w = fill(name, 128, 0.1, 1.0)
        keep.append(w)
        hip._check(L.vc_flux_bind_weight(h, name.encode(), w.data_ptr(), None, 1, 128, 128), name)
    try:
        for name, rows, cols in (("img_in", D, IN_CH), ("txt_in", D, CTX), ("time_in.in_layer", D, 256), ("time_in.out_layer", D, D),
                                 ("vector_in.in_layer", D, VEC), ("vector_in.out_layer", D, D), ("guidance_in.in_layer", D, 256),
                                 ("guidance_in.out_layer", D, D), ("final_layer.linear", OUT_CH, D)):
            bind_linear(name, rows, cols)
        for i in range(DEPTH):
            for st in ("img", "txt"):
                bind_linear(f"double_blocks.{i}.{st}_attn.qkv", 3 * D, D)
                bind_linear(f"double_blocks.{i}.{st}_attn.proj", D, D)
                bind_linear(f"double_blocks.{i}.{st}_mlp.0", MLP, D)
                bind_linear(f"double_blocks.{i}.{st}_mlp.2", D, MLP)
                bind_scale(f"double_blocks.{i}.{st}_attn.norm.query_norm.scale")
                bind_scale(f"double_blocks.{i}.{st}_attn.norm.key_norm.scale")
        for i in range(SINGLE):
            bind_linear(f"single_blocks.{i}.linear1", 3 * D + MLP, D)
            bind_linear(f"single_blocks.{i}.linear2", D, D + MLP)
            bind_scale(f"single_blocks.{i}.norm.query_norm.scale")
            bind_scale(f"single_blocks.{i}.norm.key_norm.scale")
        n_mod = L.vc_flux_mod_offset(h, None)
        assert n_mod == DEPTH * 12 * D + SINGLE * 3 * D + 2 * D
        bind_linear("modulation", n_mod, D)
        txt, y = fill("input.txt", T * CTX, 1.0), fill("input.y", VEC, 1.0)
        x, cond = fill("input.x", N * OUT_CH, 1.0).reshape(N, OUT_CH), fill("input.cond", N * (IN_CH - OUT_CH), 1.0).reshape(N, IN_CH - OUT_CH)
        r_ = np.arange(N)
        img_ids = np.stack([r_ // 12 + 1, (r_ % 12) // 6, r_ % 6], 1).astype(np.float32).copy()
        txt_ids, guidance = np.zeros((T, 3), np.float32), np.asarray([30.0], np.float32)
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))  # noqa: E731
        ws = torch.empty(L.vc_flux_workspace_bytes(h, 1, T, N, STEPS) + 256, dtype=torch.uint8, device="cuda:0")
        base = (ws.data_ptr() + 255) & ~255
        st = torch.cuda.Stream()
        s = st.cuda_stream
        inp = hip.FluxInputs(1, T, N, STEPS, txt.data_ptr(), y.data_ptr(), fp(guidance), fp(img_ids), fp(txt_ids), None, None, 0, 0)
        torch.cuda.synchronize()
        hip._check(L.vc_flux_prepare(h, C.byref(inp), base, ws.numel() - 256, s), "prepare")
        img = torch.empty(N, IN_CH, dtype=torch.bfloat16, device="cuda:0")
        fwd = torch.empty(N, OUT_CH, dtype=torch.bfloat16, device="cuda:0")
        traj = torch.empty(STEPS, N, OUT_CH, dtype=torch.bfloat16, device="cuda:0")
        hip._check(L.vc_concat_cols(x.data_ptr(), OUT_CH, cond.data_ptr(), IN_CH - OUT_CH, img.data_ptr(), N, s), "concat")
        t07 = np.asarray([0.7], np.float32)
        hip._check(L.vc_flux_forward(h, img.data_ptr(), fp(t07), 0, fwd.data_ptr(), s), "forward")
        grid = np.asarray([0.0, 0.25, 0.5, 0.75, 1.0], np.float32)
        xs = x.clone()
        hip._check(L.vc_flux_sample_euler(h, xs.data_ptr(), cond.data_ptr(), fp(grid), STEPS + 1, 1, traj.data_ptr(), s), "sample")
        torch.cuda.synchronize()
    finally:
        L.vc_flux_destroy(h)
    assert torch.isfinite(got.float()).all() and float(got.float().abs().max()) > 0
    assert torch.equal(got[0].cpu(), fwd.cpu())
    assert torch.equal(got[1].cpu(), xs.cpu())
    assert torch.equal(got[2:].cpu(), traj.cpu()) and torch.equal(traj[-1], xs)
    assert not torch.equal(traj[0], traj[1])
