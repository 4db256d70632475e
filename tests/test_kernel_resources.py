"""Scratch (spilled registers) of the kernels that must not have any, from the built library's code-object metadata (no GPU).
A regression guard: in round 5 a code path added to the shared GEMM epilogue took `convgemm_dma_kernel<8, 1, 4, ...>` from 0 to 1040 B
of scratch per lane -- small / base ES's decoder-head GEMM +35 % -- and only the bench line showed it."""
import os, re, shutil, subprocess, sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "efficientspeech_amd", "libesmi.so")
TOOLS = ["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "/opt/rocm/lib/llvm/bin/llvm-readelf"]


@pytest.mark.skipif(not os.path.exists(LIB) or not all(os.path.exists(t) for t in TOOLS) or shutil.which("objcopy") is None,
                    reason="needs the built libesmi.so and the ROCm LLVM tools")
def test_no_scratch_on_the_hot_kernels():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_resources.py"), LIB, "--all"], capture_output=True, text=True, check=True).stdout
    rows = {m.group(1): int(m.group(2)) for m in re.finditer(r"^\| `(.+?)` \| [^|]+ \| (\d+) \|", out, re.M)}
    assert len(rows) > 100, out[-400:]
    zero = [r"^enc_all16_kernel<", r"^enc_b0_16_kernel<", r"^enc_b1_16_kernel$", r"^enc_va16_kernel<", r"^mel_decoder_kernel<128, ", r"^mel_decoder_kernel<256, ", r"^enc_va64_kernel<", r"^enc_post_attn64_kernel<", r"^enc_pred128_kernel$", r"^enc_fuse128_kernel<", r"^enc_merge_q256_kernel<", r"^pwgemm_kernel<\d, 2, ",
            r"^train_conv_wgrad_mfma_kernel<", r"^convgemm_kernel<[124], ", r"^convgemm_dma_kernel<4, ", r"^convgemm_dma_kernel<8, 1, 4, (true|false), true>", r"^attn_lds_kernel<"]
    for pat in zero:
        hit = {k: v for k, v in rows.items() if re.search(pat, k)}
        assert hit, pat
        assert all(v == 0 for v in hit.values()), {k: v for k, v in hit.items() if v}
    small = {k: v for k, v in rows.items() if re.search(r"^(convgemm_dma_kernel<8|pwgemm_kernel<\d, 4)", k)}
    assert all(v <= 64 for v in small.values()), small          # (the unpacked 256-channel GEMM's 28 B)
    parked = {k: v for k, v in rows.items() if re.search(r"^enc_post_attn128_kernel$", k)}
    assert parked and all(v <= 256 for v in parked.values()), parked   # (tile 1's operand parked in scratch during tile 0's FFN pass: 196 B, once per wave)
