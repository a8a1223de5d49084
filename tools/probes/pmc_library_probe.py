"""One symmetric eigen-solve of a 640 x 640 matrix, for running under `rocprofv3 --pmc`: with argument `library` the
tridiagonal stage and the back-product go through rocSOLVER stedc / rocBLAS sgemm (SLIM_EIGH_LIBRARY_TAIL=1), with
`own` through the library's own kernels.  Tells whether counter collection fails on the vendor libraries' launches or on
this library's."""
import os
import sys

os.environ["SLIM_EIGH_LIBRARY_TAIL"] = "1" if sys.argv[1] == "library" else "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from slim_amd import ops

g = torch.Generator(device="cuda").manual_seed(0)
A = torch.randn(640, 640, generator=g, device="cuda")
A = A @ A.t()
evals, vecs, info = ops.eigh(A, top=512)
torch.cuda.synchronize()
print("eigh done, info", int(info.abs().sum()), "largest eigenvalue", float(evals[-1]))
