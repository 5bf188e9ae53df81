"""The LDS layouts of the batch kernels against the bank model of the micro-architecture guide (tools/lds_bank_check.py, CPU only):
what DESIGN.md section 5 says about conflict-free operand reads, and what it says is still open, stays true as the constants move."""
import os
import subprocess
import sys


def test_bank_model_of_the_batch_kernels():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "lds_bank_check.py")], capture_output=True, text=True, check=True).stdout
    lines = [l.strip() for l in out.splitlines()]
    pad = {int(l.split("shifted by")[1].split("B")[0]): l for l in lines if "shifted by" in l}
    assert all("consumer ds_read_b128 1 pass" in l for l in pad.values())  # the chunk mat-mul's A operands: free at every shift
    assert "producer ds_write_b128 2 pass" in pad[0] and "producer ds_write_b128 1 pass" in pad[64]
    att = [l for l in lines if l.startswith("rows ")]
    assert "2 pass" in att[0] and "1 pass" in att[1]  # batch attention: padded rows conflict under the real lane groups, the swizzle does not
    f16 = [l for l in lines if l.startswith("piece ^")]
    assert "2 pass" in f16[0] and "(f16_gemm_w8_kernel): ds_read_b128 1 pass" in f16[1]  # the fp16 GEMM's stage image: the swizzle the kernel uses is free under the real lane groups
