// rocSOLVER / rocBLAS entry points shared by the low-rank fit and the symmetric eigen-solver, resolved with dlopen at
// first use (values of the rocBLAS enums are part of its stable C ABI).
#pragma once
#include "common.hpp"
#include <map>
#include <mutex>
#include <utility>

typedef void* rb_handle;
enum {
    RB_OP_NONE = 111, RB_OP_TRANSPOSE = 112, RB_FILL_UPPER = 121, RB_FILL_LOWER = 122, RB_DIAG_NON_UNIT = 131,
    RB_EVECT_ORIGINAL = 211, RB_EVECT_TRIDIAGONAL = 212
};
typedef int (*fn_create)(rb_handle*);
typedef int (*fn_set_stream)(rb_handle, hipStream_t);
typedef int (*fn_potrf)(rb_handle, int, int, float*, int, int*);
typedef int (*fn_trtri)(rb_handle, int, int, int, float*, int, int*);
typedef int (*fn_syevd)(rb_handle, int, int, int, float*, int, float*, float*, int*);
typedef int (*fn_potrf_sb)(rb_handle, int, int, float*, int, int64_t, int*, int);
typedef int (*fn_trtri_sb)(rb_handle, int, int, int, float*, int, int64_t, int*, int);
typedef int (*fn_syevd_sb)(rb_handle, int, int, int, float*, int, int64_t, float*, int64_t, float*, int64_t, int*, int);
typedef int (*fn_stedc)(rb_handle, int, int, float*, float*, float*, int, int*);
typedef int (*fn_sgemm)(rb_handle, int, int, int, int, int, const float*, const float*, int, const float*, int, const float*,
                        float*, int);

struct Solver {
    bool ok = false;
    fn_create create = nullptr;
    fn_set_stream set_stream = nullptr;
    fn_potrf potrf = nullptr;
    fn_trtri trtri = nullptr;
    fn_syevd syevd = nullptr;
    fn_potrf_sb potrf_sb = nullptr;      // strided-batched variants: one launch chain for a whole batch
    fn_trtri_sb trtri_sb = nullptr;
    fn_syevd_sb syevd_sb = nullptr;
    fn_stedc stedc = nullptr;            // tridiagonal divide and conquer (optional: the in-library eigen-solver needs it)
    fn_sgemm sgemm = nullptr;
    std::map<std::pair<int, hipStream_t>, rb_handle> handles;  // per (device, stream): rocSOLVER keeps device workspace in the handle
    std::mutex mu;
};
extern Solver g_solver;

// handle bound to `st` on the current device (created on first use); SLIM_ERR_SOLVER when the library cannot be loaded
int solver_handle(hipStream_t st, rb_handle* out);
