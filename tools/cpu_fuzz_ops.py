#!/usr/bin/env python3
"""Seeded random sweep of the oracle's quantized mat-mul (activation quantizer + block dots, all five weight types) against the real reference's
powerserve_compute_forward_mul_mat (oracle/_ref), dev container only: random K, N, columns, activation scales, all-zero blocks; results on bits.
usage: cpu_fuzz_ops.py <seed> <seconds>"""
import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import binding as B
from powerserve_amd import synth
o=B.Oracle(); r=B.Ref(2)
rng=np.random.default_rng(int(sys.argv[1])); n=0; bad=0; t_end=time.time()+float(sys.argv[2])
while time.time()<t_end:
    t=int(rng.choice([2,8,12,13,14])); kq=t in (12,13,14)
    K=int(rng.integers(1,17))*256 if kq else int(rng.integers(1,130))*32
    N=int(rng.integers(1,200)); bs=int(rng.choice([1,2,3,5,8,17,33]))
    w=synth.random_blocks(rng,t,N,K)
    x=(rng.standard_normal((bs,K))*rng.choice([0.01,0.3,1.0,5.0,40.0],(bs,1))).astype(np.float32)
    if rng.random()<0.2: x[rng.integers(0,bs), :min(K,256)]=0
    a=o.mul_mat(t,w,K,N,x); b=r.mul_mat(t,w,K,N,x)
    if not np.array_equal(a.view(np.uint32), b.view(np.uint32)): bad+=1; print("DIFF",t,K,N,bs)
    n+=1
print(f"op sweep seed {sys.argv[1]}: {n} quantized mat-muls (Q4_0, Q8_0, Q4_K, Q5_K, Q6_K), oracle vs real reference on bits; {bad} differences")
