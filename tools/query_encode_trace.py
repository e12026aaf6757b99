"""Where a single short query's encode time goes (BERT-base bf16, fused layer stack as a captured hipGraph): host segments by
perf_counter, the graph's device time by events.  `python -m tools.query_encode_trace`"""
import json
import time

import numpy as np
import torch

from comorag_amd.embedding_model.bge import HipBGEEmbeddingModel, pool_l2norm
from comorag_amd.utils.config_utils import BaseConfig
from tools.synthetic import random_bert, synthetic_wordpiece_tokenizer


def med(f, n=50):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts) * 1e6)


def main():
    dev = torch.device("cuda", 0)
    tok, words = synthetic_wordpiece_tokenizer()
    cfg = BaseConfig(embedding_model_name="bge-base-random-init", embedding_model_dtype="bf16")
    em = HipBGEEmbeddingModel(cfg, cfg.embedding_model_name, model=random_bert("base", vocab_size=len(tok)), tokenizer=tok)
    q = " ".join(words[:12])
    for _ in range(5):
        em.batch_encode(q)
    out = {"batch_encode_us": med(lambda: em.batch_encode(q))}
    prompt = [em.embedding_config.encode_params["passage_instruction"] + q]
    out["tokenize_pad_us"] = med(lambda: em._tokenize(prompt, 512))
    inputs = em._tokenize(prompt, 512)
    out["width"] = int(inputs["input_ids"].shape[1])

    def upload():
        return {k: v.pin_memory().to(dev, non_blocking=True) for k, v in inputs.items()}
    out["pin_upload_enqueue_us"] = med(upload)
    di = upload()
    lens = inputs["attention_mask"].numpy().sum(1).astype(np.int32)
    fz = em._fused
    key = (1, out["width"], "token_type_ids" in inputs)
    out["graph_shapes"] = [list(k) for k in fz._graphs]
    ent = fz._graphs[key]
    torch.cuda.synchronize()
    out["graph_replay_enqueue_us"] = med(lambda: ent["graph"].replay())
    torch.cuda.synchronize()

    def replay_sync():
        ent["graph"].replay(); torch.cuda.synchronize()
    out["graph_replay_to_completion_us"] = med(replay_sync)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ts = []
    for _ in range(30):
        ev[0].record(); ent["graph"].replay(); ev[1].record(); torch.cuda.synchronize(); ts.append(ev[0].elapsed_time(ev[1]) * 1e3)
    out["graph_device_us"] = float(np.median(ts))
    hidden = ent["hidden"]

    def pool_sync():
        r = pool_l2norm(hidden, di["attention_mask"]); return r.float().cpu().numpy()
    out["pool_and_download_us"] = med(pool_sync)
    out["fused_call_to_numpy_us"] = med(lambda: em._forward_pool(inputs, True).float().cpu().numpy())
    # the eager stack for the same shape, device time
    lens_dev = torch.from_numpy(lens).to(dev)
    ts = []
    for _ in range(10):
        ev[0].record(); fz._stack(di["input_ids"], lens_dev, di.get("token_type_ids")); ev[1].record(); torch.cuda.synchronize(); ts.append(ev[0].elapsed_time(ev[1]) * 1e3)
    out["eager_stack_us"] = float(np.median(ts))
    print(json.dumps(out))
    em.close()


if __name__ == "__main__":
    main()
