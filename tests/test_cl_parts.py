"""ViLCo extras and CL regularisers: Adapter, adapter EMA, BiC bias layers, narration-SSL
pooling + InfoNCE, EWC / MAS penalty kernel and consolidation -- oracle vs the reference goldens
(tests/golden/cl_parts.pt, recorded from the imported reference) on the CPU, HIP path vs goldens / oracle on the GPU."""
import os

import pytest
import torch

from parity_util import HERE, cases, rel_err


def _gold():
    return torch.load(os.path.join(HERE, "golden", "cl_parts.pt"), weights_only=False)


@pytest.mark.parametrize("kind", ["ewc", "mas"])
def test_oracle_penalty_matches_reference(kind):
    from oracle import mq_oracle
    g = _gold()[kind]
    model = cases.RegToy()
    model.load_state_dict(g['state'])
    key = 'fisher' if kind == 'ewc' else 'importance'
    pen = mq_oracle.cl_penalty(list(model.named_parameters()), g['reg_params'][key], g['reg_params']['optpar'], 0.37)
    assert abs(float(pen) + g['base_loss'] - g['loss']) <= 1e-5 * abs(g['loss'])


def test_oracle_ssl_loss_matches_reference():
    from oracle import mq_oracle
    g = _gold()['ssl']
    t, v = g['text'].clone().requires_grad_(True), g['video'].clone().requires_grad_(True)
    loss = mq_oracle.masked_contrastive_loss(t, v, g['mask'], g['memory'])
    loss.backward()
    assert abs(float(loss) - g['loss']) <= 1e-5 * g['loss']
    assert rel_err(t.grad, g['dtext']) < 1e-5 and rel_err(v.grad, g['dvideo']) < 1e-5


# ------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["ewc", "mas"])
def test_penalty_kernel_matches_reference(dev, kind):
    """vilco_cl_penalty: value and the gradient it adds to p.grad = the reference's autograd result"""
    from vilco_amd.cl_methods import regularizers
    g = _gold()[kind]
    model = cases.RegToy().to(dev)
    model.load_state_dict(g['state'])
    model.reg_params = {k: [{n: v.to(dev) for n, v in d.items()} for d in lst] for k, lst in g['reg_params'].items()}
    base = cases.reg_toy_loss(model)
    base.backward()
    pen = regularizers.apply_penalty(model, 0.37, kind=kind)
    assert abs(float(base) + float(pen) - g['loss']) <= 1e-5 * abs(g['loss'])
    for n, p in model.named_parameters():
        if n in g['grads']:
            assert rel_err(p.grad, g['grads'][n]) < 1e-5, n
    # the autograd form gives the same thing
    model.zero_grad(set_to_none=True)
    loss = regularizers.get_regularized_loss(cases.reg_toy_loss(model), model, 0.37, kind=kind)
    assert abs(float(loss) - g['loss']) <= 1e-5 * abs(g['loss'])


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["ewc", "mas"])
def test_consolidation_matches_reference(dev, kind):
    from vilco_amd.cl_methods import regularizers
    want = _gold()[kind + '_update']
    model = cases.RegToy().to(dev)
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    reg = regularizers.on_task_update(cases.reg_toy_loader(dev), dev, opt, model, kind=kind)
    assert sorted(reg) == sorted(want)
    for k in want:
        assert len(reg[k]) == len(want[k]) == 1 and sorted(reg[k][0]) == sorted(want[k][0])
        for n, w in want[k][0].items():
            assert rel_err(reg[k][0][n], w) < 1e-5, (k, n)


@pytest.mark.gpu
def test_adapter_forward_backward_vs_oracle(dev):
    """Adapter.forward_tm (a Linear over the TIME axis, T -> 5T -> T/2, meta_archs.py:105-148) vs the fp64 oracle"""
    from oracle import mq_oracle
    from vilco_amd.modeling.meta_archs import Adapter
    torch.manual_seed(0)
    T, D, B = 64, 48, 2
    ad = Adapter(embed_dim=T, down_sample=5, mode='parallel', scale='null')
    with torch.no_grad():
        ad.layer[2].weight.normal_(0, 0.05)
        ad.layer[2].bias.normal_(0, 0.05)
    p64 = {'a.' + k: v.detach().double().requires_grad_(True) for k, v in ad.state_dict().items()}
    x = torch.randn(B, T, D)
    x64 = x.double().requires_grad_(True)
    want = mq_oracle.adapter(p64, 'a.', x64.permute(0, 2, 1)).permute(0, 2, 1)             # oracle is channel-first
    w = torch.randn(B, T // 2, D, dtype=torch.float64)
    (want * w).sum().backward()
    ad = ad.to(dev)
    xg = x.to(dev).requires_grad_(True)
    got = ad.forward_tm(xg)
    (got * w.float().to(dev)).sum().backward()
    assert rel_err(got, want) < 1e-4
    assert rel_err(xg.grad, x64.grad) < 1e-4
    for k, v in ad.named_parameters():
        assert rel_err(v.grad, p64['a.' + k].grad, 1e-7) < 1e-4, k


@pytest.mark.gpu
def test_adapter_ema_update(dev):
    """post_train_step: ema = 0.999 ema + 0.001 adapters over every state_dict value (meta_archs.py:702-707)"""
    from vilco_amd.modeling.meta_archs import Adapter
    from vilco_amd.utils.model_ema import ModelEmaV2
    torch.manual_seed(1)
    pets = torch.nn.ModuleList([Adapter(embed_dim=32), Adapter(embed_dim=16)]).to(dev)
    ema = ModelEmaV2(pets, decay=0.999)
    want = {k: v.detach().double().cpu().clone() for k, v in ema.module.state_dict().items()}
    for step in range(3):
        with torch.no_grad():
            for p in pets.parameters():
                p.add_(0.1 * torch.randn_like(p))
        ema.update(pets)
        for k, v in pets.state_dict().items():
            want[k] = 0.999 * want[k] + 0.001 * v.detach().double().cpu()
    for k, v in ema.module.state_dict().items():
        assert rel_err(v, want[k]) < 1e-6, k
    assert all(not p.requires_grad or True for p in ema.module.parameters())


@pytest.mark.gpu
def test_bic_bias_layers(dev):
    from oracle import mq_oracle
    from parity_util import build_hip_model, load_golden
    from vilco_amd.modeling.meta_archs import BiasLayer
    model = build_hip_model(load_golden("noxl"))
    model.list_splits = [5, 12, 22]
    model.list_bias_layers = [BiasLayer().to(dev) for _ in model.list_splits]
    with torch.no_grad():
        for i, b in enumerate(model.list_bias_layers):
            b.alpha.fill_(1.0 + 0.1 * i)
            b.beta.fill_(-0.2 * i)
    x = torch.randn(2, 40, 22, device=dev)
    got = model._bic_correct(x)
    want = mq_oracle.bic_correct(x.double().cpu(), model.list_splits, [float(b.alpha) for b in model.list_bias_layers],
                                 [float(b.beta) for b in model.list_bias_layers])
    assert rel_err(got, want) < 1e-6


@pytest.mark.gpu
def test_narration_ssl_branch_vs_oracle(dev):
    """masked mean pooling of narration tokens / pyramid features, normalisation, memory-bank update and the
    InfoNCE loss (meta_archs.py:794-811, 38-60, 1351-1372) on the device vs the oracle; the loss is also pinned to the
    reference golden through tests/golden/cl_parts.pt."""
    from oracle import mq_oracle
    from parity_util import build_hip_model, load_golden
    from vilco_amd.modeling.blocks import lens_to_mask
    from vilco_amd.modeling.meta_archs import MemoryBank
    g = _gold()['ssl']
    model = build_hip_model(load_golden("noxl"))
    D = 32
    torch.manual_seed(2)
    model.narration_ssl = True
    model.narration_encoder = torch.nn.Linear(12, D).to(dev)        # the reference hard-wires 1024 = its embd_dim
    model.memory_bank = MemoryBank(10, D, device=dev)
    B, Ts = 3, [16, 8, 4]
    feats = [torch.randn(B, T, D, device=dev) for T in Ts]                            # token-major pyramid
    lens = [torch.tensor([T, T - 3, 1], dtype=torch.int32, device=dev).clamp(max=T) for T in Ts]
    nb = torch.randn(B, 12, 5, device=dev)
    m1 = torch.tensor([[1, 1, 1, 0, 0], [1, 1, 1, 1, 1], [0, 0, 0, 0, 0]], dtype=torch.float32, device=dev)[:, None, :]
    m0 = torch.tensor([1.0, 1.0, 0.0], device=dev)
    narr, video = model._ssl_embeddings(feats, lens, (nb, m0, m1))
    p = {'narration_encoder.weight': model.narration_encoder.weight.detach().double().cpu(),
         'narration_encoder.bias': model.narration_encoder.bias.detach().double().cpu()}
    wn, wv = mq_oracle.ssl_embeddings(p, [f.double().cpu().permute(0, 2, 1) for f in feats],
                                      [lens_to_mask(l, T).cpu() for l, T in zip(lens, Ts)], nb.double().cpu(),
                                      m1.double().cpu())
    assert rel_err(narr, wn) < 1e-5 and rel_err(video, wv) < 1e-5
    # InfoNCE on the reference golden's inputs
    model.memory_bank = MemoryBank(10, 16, device=dev)
    model.memory_bank.memory.copy_(g['memory'].to(dev))
    t, v = g['text'].to(dev).requires_grad_(True), g['video'].to(dev).requires_grad_(True)
    loss = model.masked_contrastive_loss(t, v, g['mask'].to(dev))
    loss.backward()
    assert abs(float(loss) - g['loss']) <= 1e-5 * g['loss']
    assert rel_err(t.grad, g['dtext']) < 1e-5 and rel_err(v.grad, g['dvideo']) < 1e-5
    # ring-buffer update wraps around like the reference's (meta_archs.py:46-57)
    mb = MemoryBank(4, 2, device=dev)
    mb.update(torch.ones(3, 2, device=dev))
    mb.update(2 * torch.ones(3, 2, device=dev))
    assert mb.ptr == 2 and mb.memory[:, 0].tolist() == [2.0, 2.0, 1.0, 2.0]
