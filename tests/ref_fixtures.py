"""Loaders for tests/golden/ref_*.npz: outputs of the REFERENCE'S OWN CODE (src/rnnt/model.py), captured in
the build container by tests/golden/make_ref_fixtures.py.  Data only: inputs, parameters, expected outputs."""
import os

import numpy as np
import torch
import torch.nn as nn

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def joint_case(kind):
    """-> dict of float64 torch tensors: enc, dec, go, logits, grad_enc, grad_dec, params[4], grads[4]."""
    z = load(f"ref_joint_{kind}")
    assert str(z["kind"]) == kind
    names = ("mlp.0.weight", "mlp.0.bias", "mlp.2.weight", "mlp.2.bias")
    out = {k: torch.from_numpy(z[k]) for k in ("enc", "dec", "go", "logits", "grad_enc", "grad_dec")}
    out["names"] = names
    out["params"] = [torch.from_numpy(z["param." + n]) for n in names]
    out["grads"] = [torch.from_numpy(z["grad." + n]) for n in names]
    return out


class FixtureDecoder(nn.Module):
    """The layers of the reference's prediction network on the decoding path (src/decoder/decoder.py:5-44:
    embedding -> LSTM -> Linear, called as decoder(token[1,1], hidden=hidden)), built from stock torch
    modules so that the fixture's ``decoder.*`` arrays load by name."""

    def __init__(self, vocab, hidden, out):
        super().__init__()
        self.embedding = nn.Embedding(vocab, hidden, padding_idx=0)
        self.lstm = nn.LSTM(hidden, hidden, 1, batch_first=True)
        self.output_proj = nn.Linear(hidden, out)

    def forward(self, inputs, length=None, hidden=None):
        o, hidden = self.lstm(self.embedding(inputs), hidden)
        return self.output_proj(o), hidden


def recognize_case():
    """-> (enc_states [B,T,He], out_lengths, expected token lists, decoder module, joint params[4], margin)."""
    z = load("ref_recognize")
    dec = FixtureDecoder(int(z["vocab"]), int(z["dec_hidden"]), int(z["dec_out"]))
    dec.load_state_dict({k[len("decoder."):]: torch.from_numpy(v) for k, v in z.items() if k.startswith("decoder.")})
    jp = [torch.from_numpy(z["joint." + n]) for n in ("mlp.0.weight", "mlp.0.bias", "mlp.2.weight", "mlp.2.bias")]
    off = z["token_offsets"]
    tokens = [z["tokens"][off[i]:off[i + 1]].tolist() for i in range(len(off) - 1)]
    return (torch.from_numpy(z["enc_states"]), z["out_lengths"].tolist(), tokens, dec.eval(), jp,
            float(z["min_top2_margin"]))
