// (batched mat-mul lives in k_gemv.hip: gemm8_q4k_kernel shares the mat-vec's weight layout helpers)
