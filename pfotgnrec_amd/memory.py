"""Node memory store - mirror of modules/memory.py with a dense device layout.

The reference keeps ``messages: defaultdict(list)`` of ``(tensor, time)`` tuples per node; only the
last message of a node is ever consumed (``last`` aggregator, message_aggregator.py:38-55) and a
node's list holds messages of one batch only.  Here the pending state is three
HBM-resident tables written by ``pfo_tgn_update_state``:

    msg_table f32[n_nodes, 3D+Ef]   last pending raw message
    msg_time  f32[n_nodes]          its timestamp
    has_msg   u8 [n_nodes]          "a message is pending"
"""
from collections import defaultdict

import torch
from torch import nn


class Memory(nn.Module):
    def __init__(self, n_nodes, memory_dimension, input_dimension, message_dimension=None, device="cpu",
                 combination_method="sum"):
        super().__init__()
        self.n_nodes = n_nodes
        self.memory_dimension = memory_dimension
        self.input_dimension = input_dimension
        self.message_dimension = message_dimension
        self.device = torch.device(device)
        self.combination_method = combination_method
        # Parameters (requires_grad=False) exactly like modules/memory.py:28-31 so they are saved with the model
        self.memory = nn.Parameter(torch.zeros((n_nodes, memory_dimension), device=self.device), requires_grad=False)
        self.last_update = nn.Parameter(torch.zeros(n_nodes, device=self.device), requires_grad=False)
        self.msg_table = torch.zeros((n_nodes, input_dimension), device=self.device)
        self.msg_time = torch.zeros(n_nodes, device=self.device)
        self.has_msg = torch.zeros(n_nodes, dtype=torch.uint8, device=self.device)
        # host-side knowledge "some node holds a pending message": False = unknown (the TGN reads has_msg back once);
        # the native state update sets it.  Decides whether the GRU takes part in a step (TGN._attach_grads)
        self._any_msg = False
        # counts the rewrites of the tables from outside the training step (a batch prepared ahead of time holds packed copies
        # of memory rows: TGN.prefetch stamps them with this number)
        self._state_version = 0

    def __init_memory__(self):
        """Zero the memory and drop every pending message (modules/memory.py:23-33); called per epoch (main.py:153).

        In place, so that device pointers held by the native state stay valid.
        """
        with torch.no_grad():
            self.memory.zero_()
            self.last_update.zero_()
            self.msg_table.zero_()
            self.msg_time.zero_()
            self.has_msg.zero_()
        self._any_msg = False
        self._state_version += 1

    def get_memory(self, node_idxs):
        return self.memory[torch.as_tensor(node_idxs, device=self.memory.device, dtype=torch.long), :]

    def set_memory(self, node_idxs, values):
        with torch.no_grad():
            self.memory[torch.as_tensor(node_idxs, device=self.memory.device, dtype=torch.long), :] = values
        self._state_version += 1

    def get_last_update(self, node_idxs):
        return self.last_update[torch.as_tensor(node_idxs, device=self.memory.device, dtype=torch.long)]

    def backup_memory(self):
        """modules/memory.py:48-53 - third element is the pending-message state (tables instead of a dict)."""
        return (self.memory.data.clone(), self.last_update.data.clone(),
                (self.msg_table.clone(), self.msg_time.clone(), self.has_msg.clone()))

    def restore_memory(self, memory_backup):
        with torch.no_grad():
            self.memory.copy_(memory_backup[0])
            self.last_update.copy_(memory_backup[1])
            tab, t, has = memory_backup[2]
            self.msg_table.copy_(tab)
            self.msg_time.copy_(t)
            self.has_msg.copy_(has)
        self._any_msg = False
        self._state_version += 1

    def __setattr__(self, name, value):
        if name[0] == "_" and not isinstance(value, (nn.Parameter, nn.Module)):      # plain bookkeeping, written every batch
            object.__setattr__(self, name, value)
            return
        super().__setattr__(name, value)

    def detach_memory(self):
        """modules/memory.py:62-71.  Stored messages and memory never carry an autograd graph here."""
        return None

    def clear_messages(self, nodes):
        with torch.no_grad():
            self.has_msg[torch.as_tensor(nodes, device=self.has_msg.device, dtype=torch.long)] = 0
        self._any_msg = False
        self._state_version += 1          # a batch prepared ahead of time holds packed has_msg / message rows

    @property
    def messages(self):
        """Dict view ``node -> [(message, time)]`` of the pending table (inspection / tests only)."""
        out = defaultdict(list)
        has = self.has_msg.cpu().numpy()
        if has.any():
            tab, t = self.msg_table.cpu(), self.msg_time.cpu()
            for nid in has.nonzero()[0]:
                out[int(nid)] = [(tab[nid], t[nid])]
        return out
