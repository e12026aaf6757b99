import sys, json; sys.path.insert(0, '.')
import torch
from tools.bench_extras import f1_selfjoin
d = torch.device('cuda', 0)
for rep in range(2):
    r = f1_selfjoin(torch, d)
    b = r['bf16']; print({k: b[k] for k in ('threshold_search_whole_join_s','ids_download_s','one_stream_whole_join_s','us_per_pass','frac','same_neighbours_above_threshold')}, flush=True)
