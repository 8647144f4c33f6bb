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
  minilm_wide_seed5.npz/.json   second embedder pin: the "wide" style-1 weights (bell-shaped values, LayerNorm gains
                     1 +- 0.5, biases 0.1-0.2: synth.bert_tensor_specs_wide), sequences of 2 .. 512 tokens (512 =
                     max_position_embeddings, the longest input the model admits)
  scan_normal_seed4.npz  scan fixture on bell-shaped unit rows with four heavy dimensions (synth.unit_rows_normal):
                     queries, top-20 labels/distances over 20000 rows, by tests/np_oracle.py
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
    # ---- second embedder pin: wide weights, long sequences
    seed2 = 5
    w2 = synth.bert_weights(seed2, style=1)
    model2 = transformers.BertModel(cfg, add_pooling_layer=False).eval()
    res = model2.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in w2.items()}, strict=False)
    assert not res.missing_keys and not res.unexpected_keys, res
    lens2 = [2, 3, 17, 64, 65, 128, 200, 512]
    seqs2 = [synth.token_sequences(30 + i, 1, L, L)[0] for i, L in enumerate(lens2)]
    assert [len(x) for x in seqs2] == lens2
    embs2, hs_last = [], None
    with torch.no_grad():
        for i, sq in enumerate(seqs2):
            ids = torch.tensor(sq.astype(np.int64))[None]
            out = model2(input_ids=ids, attention_mask=torch.ones_like(ids),
                         token_type_ids=torch.zeros_like(ids)).last_hidden_state[0].numpy()
            if i == 2:
                hs_last = out.astype(np.float32)
            pooled = (out.sum(0) / len(sq)).astype(np.float32)
            embs2.append(synth.normalize_rows(pooled[None])[0])
    offs2 = np.concatenate([[0], np.cumsum(lens2)]).astype(np.int32)
    np.savez_compressed(os.path.join(HERE, "minilm_wide_seed5.npz"), token_ids=np.concatenate(seqs2).astype(np.uint32),
                        seq_offsets=offs2, embeddings=np.stack(embs2).astype(np.float32), hidden_states_seq2=hs_last)
    json.dump({"weight_seed": seed2, "weight_style": 1,
               "source": "transformers.BertModel hidden_act=gelu_new, no mask, no pooler; synth.bert_weights(5, style=1)",
               "transformers": transformers.__version__, "torch": torch.__version__, "lengths": lens2},
              open(os.path.join(HERE, "minilm_wide_seed5.json"), "w"), indent=1)

    # ---- scan fixture on bell-shaped rows with heavy dimensions
    ids = np.arange(1, n + 1, dtype=np.uint64)
    Xn = synth.unit_rows_normal(4, 0, n, heavy_dims=(7, 101, 213, 340))
    Qn = synth.unit_rows_normal(6, 0, 8, heavy_dims=(7, 101, 213, 340))
    Qn[3] = Xn[1234]
    labs, dists = [], []
    for q in Qn:
        l, d = NP.scan_topk(Xn, ids, q, 20)
        labs.append(l)
        dists.append(d)
    np.savez_compressed(os.path.join(HERE, "scan_normal_seed4.npz"), queries=Qn, labels=np.stack(labs), distances=np.stack(dists),
                        n_rows=np.int64(n), index_seed=np.int64(4), heavy_dims=np.array([7, 101, 213, 340]))
    print("golden fixtures written")


if __name__ == "__main__":
    main()
