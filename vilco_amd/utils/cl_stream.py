"""Task streams for the episode loop with the contract of the reference's `QILSetTask`
(MQ/libs/datasets/cl_benchmark.py:18-140) -- over clips that are already in memory.

The reference builds every task's loader from Ego4D feature files (libs/datasets/ego4d.py, outside the hot path);
what the episode loop consumes is only this contract, which `InMemoryQILStream` keeps:

  iter(stream) resets memory / task counter;  next(stream) -> (data, loader, num_next_classes) where
    data             = {class_id: [video dict, ...]} of the task's NEW classes               (cl_benchmark.py:77-79)
    loader           = batches (lists of video dicts) over  {**memory, **data}  -- replayed clips first, duplicates
                       by video 'id' dropped, 'is_memory' set on every dict               (:80-85, ego4d.py:453-458)
    num_next_classes = #classes of the following task, None after the last                 (:91-95)
  stream.memory is assigned by the driver after every task (train_cl.py:355);  stream.num_tasks.

`DistributedBatchLoader` is the sampler side of torchrun + DistributedSampler (datasets.py:24): every rank walks a
disjoint, equally long shard of the same seeded permutation, re-drawn by `sampler.set_epoch`.
"""
import random


class _Sampler:
    def __init__(self):
        self.epoch = 0

    def set_epoch(self, epoch):
        self.epoch = epoch


class DistributedBatchLoader:
    """batches of `batch_size` video dicts; shuffle with (seed, epoch); rank r of `world` takes items r, r+world, ...
    of the permutation padded (by wrapping) to a multiple of world * batch_size... only when drop_last is False."""

    def __init__(self, items, batch_size, shuffle=True, seed=0, rank=0, world=1, drop_last=True):
        self.items, self.batch_size, self.shuffle, self.seed = list(items), batch_size, shuffle, seed
        self.rank, self.world, self.drop_last = rank, world, drop_last
        self.sampler = _Sampler()

    def _order(self):
        idx = list(range(len(self.items)))
        if self.shuffle:
            random.Random(self.seed * 1000003 + self.sampler.epoch).shuffle(idx)
        per = len(idx) // self.world if self.drop_last else -(-len(idx) // self.world)
        if not self.drop_last and per * self.world > len(idx):
            idx = idx + idx[:per * self.world - len(idx)]          # DistributedSampler pads by wrapping around
        return idx[self.rank:per * self.world:self.world]

    def __len__(self):
        n = len(self._order())
        return n // self.batch_size if self.drop_last else -(-n // self.batch_size)

    def __iter__(self):
        order = self._order()
        for b in range(len(self)):
            yield [self.items[i] for i in order[b * self.batch_size:(b + 1) * self.batch_size]]


class InMemoryQILStream:
    def __init__(self, set_tasks, batch_size=2, shuffle=True, seed=0, train_enable=True, rank=0, world=1):
        """set_tasks: list (one per task) of {class_id: [video dict with 'id' / 'video_id', ...]}"""
        self.set_tasks = list(set_tasks)
        self.num_tasks = len(self.set_tasks)
        self.batch_size, self.shuffle, self.seed, self.train_enable = batch_size, shuffle, seed, train_enable
        self.rank, self.world = rank, world
        self.memory, self.current_task = {}, 0

    def __iter__(self):
        self.memory, self.current_task = {}, 0
        return self

    @staticmethod
    def _tag(data, is_memory):
        for videos in data.values():
            for v in videos:
                v['is_memory'] = is_memory
        return dict(data)

    @staticmethod
    def flatten(comp_data):
        """dataset order of ego4d.py:453-458: class by class, first occurrence of a video id wins"""
        seen, out = set(), []
        for videos in comp_data.values():
            for v in videos:
                key = v.get('id', v.get('video_id'))
                if key not in seen:
                    seen.add(key)
                    out.append(v)
        return out

    def __next__(self):
        if self.current_task >= self.num_tasks:
            raise StopIteration
        data = self.set_tasks[self.current_task]
        new = self._tag(data, False)
        comp = {**self._tag(self.memory, True), **new} if self.train_enable else new
        loader = DistributedBatchLoader(self.flatten(comp), self.batch_size, self.shuffle, self.seed, self.rank, self.world)
        self.current_task += 1
        nxt = len(self.set_tasks[self.current_task]) if self.current_task < self.num_tasks else None
        return data, loader, nxt

    def set_memory(self, memory):
        self.memory = memory
