"""GPU tests for the greedy-decoding widening: rnnt_row_argmax against
torch.argmax, and rnnt_amd.greedy_decode against the restated reference loop (oracle/decode_oracle.py)
run with the same modules on the same device.  Token sequences must be identical whenever the
reference's own top-2 margin is above rounding level (1e-5) at every step.  PARITY UNPINNED."""
import pytest
import torch
import torch.nn as nn

from oracle.decode_oracle import greedy_decode_reference

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


class Decoder(nn.Module):
    """Same interface and layers as the reference's BaseDecoder (src/decoder/decoder.py:5-44) on the
    decoding path: embedding -> LSTM -> Linear, called as decoder(token[1,1], hidden=hidden)."""

    def __init__(self, hidden, vocab, out):
        super().__init__()
        self.embedding = nn.Embedding(vocab, hidden, padding_idx=0)
        self.lstm = nn.LSTM(hidden, hidden, 1, batch_first=True)
        self.output_proj = nn.Linear(hidden, out)

    def forward(self, inputs, length=None, hidden=None):
        o, hidden = self.lstm(self.embedding(inputs), hidden)
        return self.output_proj(o), hidden


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("R,V", [(1, 1), (3, 7), (32, 4232), (5, 64), (64, 5455), (2, 70001)])
def test_row_argmax_matches_torch(dtype, R, V):
    import rnnt_amd

    g = torch.Generator(device=DEV).manual_seed(R * 131 + V)
    x = torch.randn(R, V, device=DEV, generator=g).to(dtype)
    assert torch.equal(rnnt_amd.row_argmax(x).long(), torch.argmax(x.float(), dim=1))


def test_row_argmax_ties_and_nan():
    import rnnt_amd

    x = torch.zeros(4, 300, device=DEV)
    x[0, [17, 200]] = 5.0            # tie: the first index wins
    x[1, 299] = 1.0
    x[2, 150] = float("nan")         # NaN counts as the maximum
    x[2, 10] = float("inf")
    x[3, :] = -float("inf")          # all equal: index 0
    assert rnnt_amd.row_argmax(x).tolist() == [17, 299, 150, 0]
    assert rnnt_amd.row_argmax(x).tolist() == torch.argmax(x, dim=1).tolist()


@pytest.mark.parametrize("seed,window", [(0, 32), (1, 1), (2, 7), (3, 64), (4, 5)])
def test_greedy_decode_equals_reference_loop(seed, window):
    import rnnt_amd

    torch.manual_seed(seed)
    T, He, Hd, H, V = 90, 24, 16, 48, 41
    joint = rnnt_amd.JointNet(He + Hd, H, V).to(DEV)
    with torch.no_grad():
        joint.mlp[2].bias[0] += 1.5   # blank wins on most frames, as in a trained transducer
        joint.mlp[2].weight *= 3.0
    decoder = Decoder(20, V, Hd).to(DEV)
    enc = torch.randn(T, He, device=DEV) * 2.0
    length = T - 7

    ref, margin = greedy_decode_reference(enc, length, decoder, lambda e, d: joint(e, d))
    got = rnnt_amd.greedy_decode(enc, length, decoder, joint, window=window)
    assert 0 < len(ref) < length     # the case exercises both branches
    if margin > 1e-5:
        assert got == ref
    else:                            # an argmax tie at rounding level somewhere: compare up to it
        pytest.skip(f"reference top-2 margin {margin:.2e} is at rounding level")


def test_greedy_decode_all_blank_and_argument_checks():
    import rnnt_amd

    torch.manual_seed(9)
    joint = rnnt_amd.JointNet(8 + 8, 16, 11).to(DEV)
    with torch.no_grad():
        joint.mlp[2].bias[0] += 100.0
    decoder = Decoder(8, 11, 8).to(DEV)
    enc = torch.randn(20, 8, device=DEV)
    assert rnnt_amd.greedy_decode(enc, 20, decoder, joint, window=6) == []
    assert rnnt_amd.greedy_decode(enc, 0, decoder, joint) == []
    with pytest.raises(ValueError):
        rnnt_amd.greedy_decode(enc.unsqueeze(0), 20, decoder, joint)
    with pytest.raises(RuntimeError):
        rnnt_amd.row_argmax(torch.zeros(2, 3))
