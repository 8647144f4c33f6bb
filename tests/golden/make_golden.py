"""Generate the golden fixtures under tests/golden/ — run ONCE in the build container (needs
transformers + torch CPU; the GPU box never runs this).

The reference cannot be built or imported (Rust, no toolchain), and its third-party arithmetic
(candle @ eab54e44) is not on disk, so the embedder's expected outputs come from an INDEPENDENT
implementation of the same architecture: HuggingFace transformers.BertModel configured like
src/embedding/model.rs:160-180 with hidden_act="gelu_new" (the tanh GELU candle's `gelu` computes,
model.rs:31-34), an all-ones attention mask (the reference has no mask, model.rs:335-341) and no pooler;
pooling and normalisation follow embedding_service.rs:124-136.  Weights are the seeded synthetic tensors
of dawnsearch_amd/synth.py (real all-MiniLM-L6-v2 weights are not available offline).

Outputs:
  minilm_seed3.npz   token_ids, seq_offsets, embeddings [B,384], hidden_states_seq0 [S0,384]
  minilm_seed3.json  metadata (seeds, versions, tolerances observed vs the C oracle)
  scan_seed1.npz     small scan fixture: queries, top-20 labels/distances over rows 0..19999 of stream 1,
                     computed by the numpy restatement tests/np_oracle.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from dawnsearch_amd import synth  # noqa: E402
from tests import np_oracle as NP  # noqa: E402


def main():
    import torch
    import transformers

    seed = 3
    w = synth.bert_weights(seed)
    cfg = transformers.BertConfig(**{**synth.MINILM_CONFIG, "hidden_act": "gelu_new"})
    model = transformers.BertModel(cfg, add_pooling_layer=False).eval()
    res = model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in w.items()}, strict=False)
    assert not res.missing_keys and not res.unexpected_keys, res
    seqs = synth.token_sequences(11, 12, 4, 32) + synth.token_sequences(12, 4, 100, 128)
    embs, hs0 = [], None
    with torch.no_grad():
        for i, s in enumerate(seqs):
            ids = torch.tensor(s.astype(np.int64))[None]
            out = model(input_ids=ids, attention_mask=torch.ones_like(ids),
                        token_type_ids=torch.zeros_like(ids)).last_hidden_state[0].numpy()
            if i == 0:
                hs0 = out.astype(np.float32)
            pooled = (out.sum(0) / len(s)).astype(np.float32)
            embs.append(synth.normalize_rows(pooled[None])[0])
    offs = np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).astype(np.int32)
    np.savez_compressed(os.path.join(HERE, "minilm_seed3.npz"), token_ids=np.concatenate(seqs).astype(np.uint32),
                        seq_offsets=offs, embeddings=np.stack(embs).astype(np.float32), hidden_states_seq0=hs0)
    json.dump({"weight_seed": seed, "source": "transformers.BertModel hidden_act=gelu_new, no mask, no pooler",
               "transformers": transformers.__version__, "torch": torch.__version__,
               "n_sequences": len(seqs), "lengths": [int(len(s)) for s in seqs]},
              open(os.path.join(HERE, "minilm_seed3.json"), "w"), indent=1)

    n = 20000
    X = synth.unit_rows(1, 0, n)
    ids = np.arange(1, n + 1, dtype=np.uint64)
    Q = np.concatenate([synth.unit_rows(2, 0, 6), synth.planted_queries(1, [0, 777, n - 1], 4)])
    labs, dists = [], []
    for q in Q:
        l, d = NP.scan_topk(X, ids, q, 20)
        labs.append(l)
        dists.append(d)
    np.savez_compressed(os.path.join(HERE, "scan_seed1.npz"), queries=Q, labels=np.stack(labs), distances=np.stack(dists),
                        n_rows=np.int64(n), index_seed=np.int64(1))
    print("golden fixtures written")


if __name__ == "__main__":
    main()
