// Symmetric eigen-solver with one launch per column of the tridiagonalisation (eigh.hip).
#pragma once
#include "common.hpp"

#define SLIM_EIGH_MAX_N 2048

// true when the solver can take an n x n problem (size limits, rocSOLVER stedc and rocBLAS sgemm present)
bool slim_eigh_supported(int n);
int64_t slim_eigh_ws_bytes(int n, int batch);
// G [batch][n, n] symmetric fp32 (destroyed).  On return evals [batch][n] ascending and rows n-top .. n-1 of each G
// hold the eigenvectors of the `top` largest eigenvalues (row n-1 the largest); info[b * info_stride] = solver status.
int slim_eigh_launch(float* G, int n, int batch, int top, float* evals, int* info, int info_stride, void* ws,
                     int64_t ws_bytes, hipStream_t st);
