"""detzero_det.datasets (detection/detzero_det/datasets/__init__.py:1-90): ``build_dataloader`` and the no-shuffle
strided ``DistributedSampler`` of distributed evaluation."""
import torch
from torch.utils.data import DataLoader, Subset
from torch.utils.data import DistributedSampler as _DistributedSampler

from detzero_amd import frame_parallel
from detzero_amd.lib import DetZeroHipError
from detzero_amd.waymo_dataset import WaymoDetectionDataset

__all__ = {'WaymoDetectionDataset': WaymoDetectionDataset}


class DistributedSampler(_DistributedSampler):
    """datasets/__init__.py:16-36: pad by wrap-around to a multiple of the world size, rank r takes indices[r::W]."""

    def __init__(self, dataset, num_replicas=None, rank=None, shuffle=True):
        super().__init__(dataset, num_replicas=num_replicas, rank=rank)
        self.shuffle = shuffle

    def __iter__(self):
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.epoch)
            order = torch.randperm(len(self.dataset), generator=g).tolist()
            order += order[:(self.total_size - len(order))]
            return iter(order[self.rank:self.total_size:self.num_replicas])
        return iter(frame_parallel.shard_indices(len(self.dataset), self.rank, self.num_replicas))


def build_dataloader(dataset_cfg, class_names, batch_size, dist, root_path=None, workers=4, logger=None, training=True,
                     merge_all_iters_to_one_epoch=False, total_epochs=0, length=0):
    """datasets/__init__.py:39-90.  Frames are assembled and voxelized on the device, which forked DataLoader workers cannot
    own, so the loader runs in-process (workers is accepted and ignored) and pin_memory is off (tensors are already in HBM)."""
    if training:
        raise DetZeroHipError('build_dataloader: the training path is out of scope of the HIP backend')
    dataset = __all__[dataset_cfg.DATASET](dataset_cfg=dataset_cfg, class_names=class_names, root_path=root_path,
                                          training=training, logger=logger)
    if dist:
        from detzero_utils import common_utils
        rank, world_size = common_utils.get_dist_info()
        sampler = DistributedSampler(dataset, world_size, rank, shuffle=False)
    else:
        sampler = None
    new_dataset = Subset(dataset, torch.arange(len(dataset))[:length]) if length > 0 else dataset
    dataloader = DataLoader(new_dataset, batch_size=batch_size, pin_memory=False, num_workers=0, shuffle=False,
                            collate_fn=dataset.collate_batch, drop_last=False, sampler=sampler, timeout=0)
    return dataset, dataloader, sampler
