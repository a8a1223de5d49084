import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_fullsize_gpu as T
from slim_amd import ops
import slim_amd.lora as L
N, K, rank = [int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (4096, 4096, 512))]
E, sr = T._error_matrix(ops, N, K, 100 + N % 97 + K % 89)
outl, spread = L.outlier_columns(sr)
sv = torch.linalg.svdvals(E.float())
U, Ut, Lf, info = L.fit_error_matrix(E, ops.transpose(E), rank, None, outl, spread)
print("first fit: report words", info[-3:].tolist(), "excess", T._excess(E, U[:, :rank], Lf[:, :rank], rank, sv)[0])
hint = L.level_hint_from_status(int(info[-2]), int(info[-3]))
print("hint level", L.hint_level(hint), "found", (hint >> 8) & 255, "ratio", (hint >> 16) / 2048, "plan", L.filter_plan([hint]))
for rounds in (3, 2):
    for margin in (0.9, 0.8):
        L.FIT_FILTER_ROUNDS, L.FIT_FILTER_MARGIN = rounds, margin
        U2, _, L2, info2 = L.fit_error_matrix(E, ops.transpose(E), rank, None, outl, spread, hint=hint)
        print("rounds", rounds, "margin", margin, "report words", info2[-3:].tolist(), "status", int(info2[:-3].abs().sum()),
              "excess", T._excess(E, U2[:, :rank], L2[:, :rank], rank, sv)[0])
