"""
ORACLE — test infrastructure only.  NOT part of the product path.

Restatement of the reference's CALLERS of the hot path — the code on either side of ``gnn.mpnn.GGNN`` that
BASELINE.json's north_star says must keep working "unchanged" (file:line relative to ``/root/reference/graphinvent/``):

* ``WorkflowOracle``: ``Workflow.__init__`` (the fields the training path reads, Workflow.py:45-79), ``get_dataloader``
  (:120-141), ``define_model_and_optimizer`` (the plain training branch, :236-261), ``create_model`` (:265-292),
  ``train_epoch`` (:766-798), ``validation_epoch`` (:800-831), ``loss`` (:833-860);
* ``GeneratorOracle``: ``GraphGenerator.__init__`` / ``initialize_graph_batch`` / ``allocate_graph_tensors`` /
  ``build_graphs`` / ``apply_actions`` / ``copy_terminated_graphs`` / ``reset_graphs`` (GraphGenerator.py:27-43,
  99-161, 163-209, 211-338, 340-387, 389-428, 430-465), with ``get_actions`` taken from ``oracle.sampler_oracle``.

Why it exists: the reference checkout is not on the GPU box, and its ``Workflow.py`` / ``GraphGenerator.py`` may not
be copied into this repository — so the ``-m gpu`` tests cannot import them.  These classes import the model and the
loader exactly the way the reference does (``import gnn.mpnn`` / ``from BlockDatasetLoader import ...`` resolved through
``sys.path``), so with ``graphinvent_amd/`` first on the path they drive the MI355X drop-ins, with
``/root/reference/graphinvent`` first the reference's own modules.

Parity pinning: ``tests/golden/make_golden_callers.py`` runs the UNMODIFIED ``Workflow`` / ``GraphGenerator`` methods in
the build container (reference ``gnn`` on CPU, stub modules only for rdkit / h5py / tensorboard / util / Analyzer …),
runs these restatements on the same inputs and asserts bit-identical losses, weights and generated graphs before it
writes ``tests/golden/golden_workflow.npz`` / ``golden_generator.npz``; ``tests/test_callers_cpu.py`` repeats that check
whenever the reference checkout is present.  The one random element of the generation loop,
``torch.distributions.Multinomial(1, probs).sample()`` (GraphGenerator.py:533-537), is a parameter here (``draw``) and is
pinned in the reference run to the same inverse-CDF draw on a seeded uniform stream (``InverseCdfDraws``).
"""
from __future__ import annotations

import numpy as np
import torch

from oracle import sampler_oracle as SO


# --------------------------------------------------------------------------------------------------------------
class WorkflowOracle:
    def __init__(self, constants):                                        # Workflow.py:45-79
        self.constants = constants
        self.test_h5_path = constants.test_set[:-3] + "h5"                # :51-53
        self.train_h5_path = constants.training_set[:-3] + "h5"
        self.valid_h5_path = constants.validation_set[:-3] + "h5"
        self.optimizer = self.scheduler = self.model = None
        self.current_epoch = self.restart_epoch = None
        self.train_dataloader = self.valid_dataloader = None

    def get_dataloader(self, hdf_path, data_description=None):            # :120-141
        from BlockDatasetLoader import BlockDataLoader, HDFDataset        # (:19: resolved through sys.path)
        dataset = HDFDataset(hdf_path)
        return BlockDataLoader(dataset=dataset, batch_size=self.constants.batch_size,
                               block_size=self.constants.block_size, shuffle=True,
                               n_workers=self.constants.n_workers, pin_memory=True)

    def create_model(self):                                               # :265-292
        import gnn.mpnn                                                   # (:23)
        name = self.constants.model
        if name == "GGNN":
            net = gnn.mpnn.GGNN(constants=self.constants)                 # :280-281
        elif name == "AttGGNN":
            net = gnn.mpnn.AttentionGGNN(constants=self.constants)        # :282-283
        else:
            raise ValueError("Invalid model entered.")                    # :287
        if self.constants.device == "cuda":
            net = net.to("cuda", non_blocking=True)                       # :289-290
        return net

    def define_model_and_optimizer(self):                                 # :236-261 (not fine-tune, not restart)
        self.restart_epoch = 0
        self.model = self.create_model()
        self.optimizer = torch.optim.Adam(params=self.model.parameters(), lr=self.constants.init_lr)
        start_epoch = self.restart_epoch + 1
        end_epoch = start_epoch + self.constants.epochs
        max_allowable_lr = self.constants.max_rel_lr * self.constants.init_lr
        n_batches = len(self.train_dataloader)
        self.scheduler = torch.optim.lr_scheduler.OneCycleLR(optimizer=self.optimizer, max_lr=max_allowable_lr,
                                                             steps_per_epoch=n_batches, epochs=self.constants.epochs)
        return start_epoch, end_epoch

    def train_epoch(self):                                                # :766-798
        training_loss_tensor = torch.zeros(len(self.train_dataloader), device=self.constants.device)
        self.model.train()
        for batch_idx, batch in enumerate(self.train_dataloader):
            if self.constants.device == "cuda":
                batch = [b.to("cuda", non_blocking=True) for b in batch]
            nodes, edges, target_output = batch
            output = self.model(nodes, edges)
            self.model.zero_grad()
            self.optimizer.zero_grad()
            batch_loss = self.loss(output=output, target_output=target_output)
            training_loss_tensor[batch_idx] = batch_loss
            batch_loss.backward()
            self.optimizer.step()
            self.scheduler.step()
        return torch.mean(training_loss_tensor)

    def validation_epoch(self):                                           # :800-831
        validation_loss_tensor = torch.zeros(len(self.valid_dataloader), device=self.constants.device)
        self.model.eval()
        with torch.no_grad():
            for batch_idx, batch in enumerate(self.valid_dataloader):
                if self.constants.device == "cuda":
                    batch = [b.to("cuda", non_blocking=True) for b in batch]
                nodes, edges, target_output = batch
                output = self.model(nodes, edges)
                validation_loss_tensor[batch_idx] = self.loss(output=output, target_output=target_output)
        return torch.mean(validation_loss_tensor)

    def loss(self, output, target_output):                                # :833-860
        output = torch.nn.LogSoftmax(dim=1)(output)
        target_output = target_output / torch.sum(target_output, dim=1, keepdim=True)
        return torch.nn.KLDivLoss(reduction="batchmean")(target=target_output, input=output)


# --------------------------------------------------------------------------------------------------------------
class InverseCdfDraws:
    """The generation loop's only random element as a seeded, implementation-independent draw: round r uses the
    uniforms ``u[r]`` and takes, per graph, the first action whose cumulative probability (float64) exceeds
    u * total — what ``gi_sample_actions`` does on the device.  ``margin`` records how close any draw came to a
    boundary of its CDF interval (a HIP forward that differs from the CPU one by 1e-6 picks the same action whenever
    the margin is larger)."""

    def __init__(self, seed: int, batch_size: int, max_rounds: int = 4096):
        self.u = np.random.default_rng(seed).random((max_rounds, batch_size))
        self.round = 0
        self.margin = 1.0

    def __call__(self, apds: np.ndarray) -> np.ndarray:
        u = self.u[self.round]
        self.round += 1
        cdf = np.cumsum(apds.astype(np.float64), axis=1)
        target = u * cdf[:, -1]
        idx = np.minimum((cdf <= target[:, None]).sum(axis=1), apds.shape[1] - 1)
        rows = np.arange(apds.shape[0])
        hi = cdf[rows, idx] - target
        lo = target - np.where(idx > 0, cdf[rows, np.maximum(idx - 1, 0)], 0.0)
        self.margin = float(min(self.margin, hi.min(), lo.min()))
        return idx


class GeneratorOracle:
    def __init__(self, model, batch_size, constants, draw):               # GraphGenerator.py:27-43
        self.batch_size, self.model, self.c, self.draw = batch_size, model, constants, draw
        self.initialize_graph_batch()
        self.allocate_graph_tensors()

    def initialize_graph_batch(self):                                     # :389-428
        c, B = self.c, self.batch_size
        self.nodes = torch.zeros([B] + list(c.dim_nodes), dtype=torch.float32, device=c.device)
        self.edges = torch.zeros([B] + list(c.dim_edges), dtype=torch.float32, device=c.device)
        self.n_nodes = torch.zeros([B], dtype=torch.int8, device=c.device)
        self.nodes[0] = torch.ones([1] + list(c.dim_nodes), device=c.device)    # the dummy non-empty graph (:424-427)
        self.edges[0, 0, 0, 0] = 1
        self.n_nodes[0] = 1

    def allocate_graph_tensors(self):                                     # :163-209
        c, B = self.c, self.batch_size
        n_allocate = B * 2
        self.generated_nodes = torch.zeros((n_allocate, *c.dim_nodes), dtype=torch.float32, device=c.device)
        self.generated_edges = torch.zeros((n_allocate, *c.dim_edges), dtype=torch.float32, device=c.device)
        self.generated_n_nodes = torch.zeros(n_allocate, dtype=torch.int8, device=c.device)
        self.likelihoods = torch.zeros((B, c.max_n_nodes * 2), device=c.device)
        self.generated_likelihoods = torch.zeros((n_allocate, c.max_n_nodes * 2), device=c.device)
        self.properly_terminated = torch.zeros(n_allocate, dtype=torch.int8, device=c.device)

    def get_actions(self, apds):                                          # :467-570 through oracle.sampler_oracle
        a = apds.detach().cpu().numpy()
        out = SO.get_actions(a, self.draw(a), self.n_nodes.cpu().numpy(), self.edges.cpu().numpy(),
                             self.c.dim_f_add, self.c.dim_f_conn)
        dev = self.c.device
        t = lambda x: torch.as_tensor(np.ascontiguousarray(x)).to(dev)
        return (tuple(t(x) for x in out["add"]), tuple(t(x) for x in out["conn"]), t(out["term"]),
                t(out["invalid"]), t(out["likelihoods"]))

    def build_graphs(self):                                               # :99-161
        softmax = torch.nn.Softmax(dim=1)
        n_generated_so_far = 0
        generation_round = 0
        while n_generated_so_far < self.batch_size:
            apd = softmax(self.model(self.nodes, self.edges))             # :121
            add, conn, term, invalid, likelihoods_just_sampled = self.get_actions(apd)
            self.properly_terminated[n_generated_so_far:(n_generated_so_far + len(term))] = 1    # :127
            termination_idc = torch.cat((term, invalid))
            termination_idc = termination_idc[termination_idc != 0]       # never the dummy graph (:133)
            n_generated_so_far = self.copy_terminated_graphs(termination_idc, n_generated_so_far, generation_round,
                                                             likelihoods_just_sampled)
            self.apply_actions(add, conn, generation_round, likelihoods_just_sampled)
            self.reset_graphs(termination_idc)
            generation_round += 1
        return n_generated_so_far

    def apply_actions(self, add, conn, generation_round, likelihoods_sampled):   # :211-338
        c = self.c
        add = [idx.long() for idx in add]
        n_node_features = [c.n_atom_types, c.n_formal_charge, c.n_imp_H, c.n_chirality]
        if not c.use_explicit_H and not c.ignore_H:                       # :264-285
            if c.use_chirality:
                batch, bond_to, atom_type, charge, imp_h, chirality, bond_type, bond_from = add
                self.nodes[batch, bond_from, chirality + sum(n_node_features[0:3])] = 1
            else:
                batch, bond_to, atom_type, charge, imp_h, bond_type, bond_from = add
            self.nodes[batch, bond_from, imp_h + sum(n_node_features[0:2])] = 1
        elif c.use_chirality:                                             # :286-294
            batch, bond_to, atom_type, charge, chirality, bond_type, bond_from = add
            self.nodes[batch, bond_from, chirality + sum(n_node_features[0:2])] = 1
        else:                                                             # :295-301
            batch, bond_to, atom_type, charge, bond_type, bond_from = add
        self.nodes[batch, bond_from, atom_type] = 1
        self.nodes[batch, bond_from, charge + n_node_features[0]] = 1
        keep = torch.nonzero(self.n_nodes[batch] != 0)                    # no dummy edge for a first atom (:304-307)
        b_m, to_m, from_m, type_m = batch[keep], bond_to[keep], bond_from[keep], bond_type[keep]
        self.edges[b_m, to_m, from_m, type_m] = 1                         # :310-311
        self.edges[b_m, from_m, to_m, type_m] = 1
        self.n_nodes[batch] += 1                                          # :314
        self.likelihoods[batch, generation_round] = likelihoods_sampled[batch]
        conn = [idx.long() for idx in conn]                               # :319-337
        batch, bond_to, bond_type, bond_from = conn
        self.edges[batch, bond_from, bond_to, bond_type] = 1
        self.edges[batch, bond_to, bond_from, bond_type] = 1
        self.likelihoods[batch, generation_round] = likelihoods_sampled[batch]

    def copy_terminated_graphs(self, terminate_idc, n_graphs_generated, generation_round, likelihoods_sampled):
        self.likelihoods[terminate_idc, generation_round] = likelihoods_sampled[terminate_idc]   # :366
        n_done_graphs = len(terminate_idc)
        begin_idx, end_idx = n_graphs_generated, n_graphs_generated + n_done_graphs              # :377-378
        self.generated_nodes[begin_idx:end_idx] = self.nodes[terminate_idc]
        self.generated_edges[begin_idx:end_idx] = self.edges[terminate_idc]
        self.generated_n_nodes[begin_idx:end_idx] = self.n_nodes[terminate_idc]
        self.generated_likelihoods[begin_idx:end_idx] = self.likelihoods[terminate_idc]
        return n_graphs_generated + n_done_graphs

    def reset_graphs(self, idc):                                          # :430-465
        if len(idc) > 0:
            self.nodes[idc] = 0
            self.edges[idc] = 0
            self.n_nodes[idc] = 0
            self.likelihoods[idc] = 0
        self.nodes[0] = torch.ones([1] + list(self.c.dim_nodes), device=self.c.device)
        self.edges[0, 0, 0, 0] = 1
        self.n_nodes[0] = 1
