"""Multi-dataset batching for joint pre-training (ponder/datasets/dataloader.py:25-117,
ponder/datasets/defaults.py:143-179).

A batch never mixes datasets - every batch carries ONE ``condition`` so the model can pick that
dataset's scene box / class table / normalisation statistics.  The sub-datasets' ``loop`` values
are read as mixing ratios: the stream is ``ratio[0]`` batches of dataset 0, ``ratio[1]`` of dataset
1, ... repeated; the epoch ends when dataset 0 (the main one, looped ``ConcatDataset.loop`` times)
is exhausted, the other datasets restart as often as needed.
"""
from functools import partial

import numpy as np
import torch
import torch.utils.data

from ..utils import comm
from .collate import loader_collate


class ConcatDataset(torch.utils.data.Dataset):
    """Index-concatenation of already built datasets."""

    def __init__(self, datasets, loop=1):
        self.datasets, self.loop = list(datasets), loop
        self.edges = np.cumsum([0] + [len(d) for d in self.datasets])

    def locate(self, idx):
        idx = idx % int(self.edges[-1])
        which = int(np.searchsorted(self.edges, idx, side="right") - 1)
        return which, idx - int(self.edges[which])

    def __getitem__(self, idx):
        which, local = self.locate(idx)
        return self.datasets[which][local]

    def __len__(self):
        return int(self.edges[-1]) * self.loop


def _seed_worker(worker_id, base):
    """ponder/datasets/dataloader.py:108-117: python, numpy and torch streams of one worker."""
    import random

    seed = base + worker_id
    random.seed(seed)
    np.random.seed(seed % (2 ** 32))
    torch.manual_seed(seed)


class _EpochSampler:
    """``set_epoch`` fan-out to the distributed samplers of the sub-loaders."""

    def __init__(self, loaders):
        self.loaders = loaders

    def set_epoch(self, epoch):
        for dl in self.loaders:
            if isinstance(dl.sampler, torch.utils.data.distributed.DistributedSampler):
                dl.sampler.set_epoch(epoch)


class MultiDatasetDataloader:
    def __init__(self, concat_dataset, batch_size_per_gpu, num_worker_per_gpu, mix_prob=0,
                 seed=None, max_point=-1, default_collate=None):
        self.datasets = concat_dataset.datasets
        self.ratios = [int(getattr(d, "loop", 1)) for d in self.datasets]
        for d in self.datasets:  # the original loops served as ratios
            d.loop = 1
        self.datasets[0].loop = concat_dataset.loop
        world, rank = comm.get_world_size(), comm.get_rank()
        self.dataloaders = []
        for k, d in enumerate(self.datasets):
            sampler = torch.utils.data.distributed.DistributedSampler(d) if world > 1 else None
            # every sub-loader collates with the point budget / Mix3D settings of the config
            # (reference :67-80)
            collate = loader_collate(d, mix_prob=mix_prob, max_point=max_point)
            init = None
            if seed is not None:  # distinct stream per (rank, dataset, worker), reference :108-117
                nw = num_worker_per_gpu // len(self.datasets)
                init = partial(_seed_worker,
                               base=nw * len(self.datasets) * rank + nw * k + seed)
            self.dataloaders.append(torch.utils.data.DataLoader(
                d, batch_size=batch_size_per_gpu, shuffle=sampler is None,
                num_workers=num_worker_per_gpu, sampler=sampler, collate_fn=collate,
                pin_memory=torch.cuda.is_available(), worker_init_fn=init, drop_last=True,
                persistent_workers=num_worker_per_gpu > 0))
        self.sampler = _EpochSampler(self.dataloaders)

    def __iter__(self):
        its = [iter(dl) for dl in self.dataloaders]
        while True:
            for k, ratio in enumerate(self.ratios):
                for _ in range(ratio):
                    try:
                        batch = next(its[k])
                    except StopIteration:
                        if k == 0:
                            return
                        its[k] = iter(self.dataloaders[k])
                        batch = next(its[k])
                    yield batch

    def __len__(self):
        main = len(self.dataloaders[0])
        return main // self.ratios[0] * sum(self.ratios) + main % self.ratios[0]
