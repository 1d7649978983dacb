"""TEST INFRASTRUCTURE ONLY - one WGAN-GP training step of the reference's loop on the CPU,
assembled from the CPU checker's pieces (train.py:236-299 of the reference).

Used by tests (loss parity of gan_amd.train.Trainer.step) and as the ``cpu_baseline`` leg of
bench.py.  Randomness (z vectors, the interpolation alpha) is passed in so that both sides see
the same draws.  Parity status: generator / constraint pieces PINNED by tests/golden; the critic
follows the restated GATConv (PARITY UNPINNED, see oracle/model.py).
"""
from __future__ import annotations

from typing import Dict, List

import torch

from gan_amd.data import Batch, Data
from gan_amd.dataset import pad_data
from . import cpu, model


def _sanitize(b: Data):
    ei, ea = model.sanitize_edges(b.edge_index, b.edge_attr)
    return ei, ea


class CpuTrainer:
    def __init__(self, gen_state: Dict[str, torch.Tensor], d_params: Dict[str, torch.Tensor], candidate_k: int,
                 max_nodes: int, max_edges: int, lr: float = 1e-4, gp_lambda: float = 10.0,
                 lambda_cons: float = 1e-2, n_critic: int = 3, segmented_readout: bool = False):
        # ``segmented_readout``: the level readout without its loop over levels (model.level_readout_segmented),
        # for graphs of 10^5 nodes, where the loop's backward pass takes minutes
        self.readout = model.level_readout_segmented if segmented_readout else None
        self.gs = {k: v.detach().clone() for k, v in gen_state.items()}
        self.dp = {k: v.detach().clone().requires_grad_(True) for k, v in d_params.items()}
        self.K, self.max_nodes, self.max_edges = candidate_k, max_nodes, max_edges
        self.gp_lambda, self.lambda_cons, self.n_critic = gp_lambda, lambda_cons, n_critic
        self.opt = torch.optim.Adam(list(self.dp.values()), lr=lr, betas=(0.5, 0.9))

    def _fake(self, tpl: Data, z: torch.Tensor) -> Data:
        out = model.generator_forward_canonical(self.gs, tpl.x, getattr(tpl, "node_depth", None), z, self.K)
        nd = getattr(tpl, "node_depth", None)
        return Data(x=out["x"], edge_index=out["edge_index"], edge_attr=out["edge_attr"],
                    node_depth=None if nd is None else nd.clone())

    def _score(self, b: Data) -> torch.Tensor:
        ei, ea = _sanitize(b)
        return model.discriminator_forward(self.dp, b.x, ei, ea, b.node_depth, readout=self.readout)

    def step(self, real_padded: Batch, templates: List[Data], zs_critic: List[torch.Tensor],
             zs_gen: List[torch.Tensor], alphas: List[torch.Tensor], validate: bool = True,
             raw_critic: List[Data] = None, raw_generator: List[Data] = None) -> Dict[str, float]:
        """``raw_critic`` / ``raw_generator``: raw graphs to use in place of this class's own generator pass (a
        replay of a recorded run of the reference feeds in the graphs that run's generator emitted)."""
        raw = [self._fake(t, z) for t, z in zip(templates, zs_critic)] if raw_critic is None else raw_critic
        corr = [cpu.enforce_aig_constraints(r) for r in raw]
        fake_padded = Batch.from_data_list([pad_data(c, self.max_nodes, self.max_edges) for c in corr])
        m: Dict[str, float] = {}
        for it in range(self.n_critic):
            self.opt.zero_grad()
            d_real = self._score(real_padded).mean()
            d_fake = self._score(fake_padded).mean()
            gp, gn = model.gradient_penalty(self.dp, real_padded.x, fake_padded.x, real_padded.node_depth, alphas[it],
                                              readout=self.readout)
            loss_d = (d_fake - d_real) + self.gp_lambda * gp
            loss_d.backward()
            self.opt.step()
            m.update(D_real=float(d_real.detach()), D_fake=float(d_fake.detach()), gap=float((d_real - d_fake).detach()), gp=float(gp.detach()),
                     grad_norm=float(gn), loss_D=float(loss_d.detach()))
        raw_gen = [self._fake(t, z) for t, z in zip(templates, zs_gen)] if raw_generator is None else raw_generator
        raw_batch = Batch.from_data_list([pad_data(r, self.max_nodes, self.max_edges) for r in raw_gen])
        self.opt.zero_grad()
        loss_adv = -self._score(raw_batch).mean()
        pen = torch.stack([cpu.aig_constraint_penalties(r)["total"] for r in raw_gen]).mean()
        if validate:
            scale = torch.tensor([10.0 if cpu.validate_strict_aig(r) else 1.0 for r in raw_gen]).mean()
            pen = pen * scale
        loss_g = loss_adv + self.lambda_cons * pen
        loss_g.backward()      # reaches only the critic's parameters, as in the reference
        m.update(loss_G=float(loss_g.detach()), pen=float(pen.detach()))
        # edges of the two critic inputs: padding columns out, and with them the self loops that batching
        # makes of the padding of every graph but the first
        m["edges"] = sum(int(((b.edge_index.min(dim=0).values >= 0) & (b.edge_index[0] != b.edge_index[1])).sum())
                         for b in (real_padded, fake_padded))
        return m
