"""BASELINE.json configs[1]: MCL EfficientNet-B0, 448x448, batch 16 on one MI355X (train_mcl.py:153-199 on B0).

Round 2 had no GPU test at this size (every B0 test ran at 64-96 px).  Three checks:
  * the reference's OWN loop body at B0 / 448x448 (tests/golden/step_b0_448_ep4.npz, written by oracle/gen_golden.py
    --config2 from train_mcl.py's AST; 4 of the 16 rows so the fixture stays small): forward tensors, the seven loss terms,
    gradient summaries, the Adam update, BatchNorm buffers - through the same assertions as the other step fixtures;
  * at the full batch of 16, where the CPU oracle is not the comparator, the conv -> train-mode-BN scale-invariance identity
    over every conv -> BN pair on one full mcl_step (tests/test_gpu_fullsize.py explains the identity);
  * the oracle on a batch-16 forward: logits / embedding of the B0 network against oracle/mcl_oracle.py at 448x448."""
import numpy as np
import pytest
import torch

import golden_util as gu
from muscle_amd import synth
from test_gpu_b7_golden import test_train_forward_values_vs_reference_step as _forward_vs_reference
from test_gpu_fullsize import scale_invariance_identity
from test_gpu_model import build, close, DEV, T, test_mcl_step_phase1_golden as _step_vs_reference

pytestmark = [pytest.mark.gpu, pytest.mark.both_arith]
FIX = "step_b0_448_ep4.npz"
FIX1 = "step_b0_224_n2_ep4.npz"          # BASELINE.json configs[0]: B0, 2 images, 224x224, one step (oracle/gen_golden.py --config1)


def test_config1_b0_224_n2_train_forward_vs_reference():
    """configs[0]'s own shape on the GPU: the reference's loop body on EfficientNet-B0, 2 synthetic 224x224 images, 21 classes
    (train_mcl.py:153-199) - forward tensors against the fixture."""
    _forward_vs_reference(FIX1)


@pytest.mark.parametrize("imc_sync", [False, True])
def test_config1_b0_224_n2_step_vs_reference(imc_sync):
    """... and the whole step: seven loss terms (IMC falls through to the Python float 0.0 with two images), gradient summaries, the
    set of grad-less parameters, the Adam update, BatchNorm buffers."""
    G = gu.load(FIX1)
    assert not bool(G["loss_is_tensor"][0]) and float(G["losses"][4]) == 0.0
    _step_vs_reference(FIX1, imc_sync)


def test_b0_448_train_forward_vs_reference():
    _forward_vs_reference(FIX)


@pytest.mark.parametrize("imc_sync", [False, True])
def test_b0_448_step_vs_reference(imc_sync):
    G = gu.load(FIX)
    assert bool(G["loss_is_tensor"][0]) and float(G["losses"][4]) > 0      # IMC is active in this fixture
    _step_vs_reference(FIX, imc_sync)


def test_b0_448_bs16_scale_invariance_identity():
    scale_invariance_identity("efficientnet-b0", 16, 448, 5000)


def test_b0_448_bs16_forward_vs_oracle():
    """Batch 16 at 448x448 against the CPU oracle (forward only: ~10 s of CPU): every tapped feature map, emb and logits."""
    from oracle import mcl_oracle as O
    name, n, size, seed = "efficientnet-b0", 16, 448, 9
    cfg, sd, model = build(name, seed)
    x = T(synth.normal(seed, "x", (n, 3, size, size)).astype(np.float32))
    du = gu.drop_draws(cfg, n, 5)
    model.train()
    with torch.no_grad():
        emb, logits = model(x.to(DEV), cam="logits", drop_u={k: v.to(DEV) for k, v in du.items()})
    net = O.OracleNet(name, sd)
    net.training = True
    with torch.no_grad():
        oemb, ologits = net.forward(x, cam="logits", drop_u=du)
    close(emb, oemb, 5e-4); close(logits, ologits, 5e-4)
