"""
One training step (forward + loss + backward) captured in a HIP graph.

The reference's batch of 12 windows is launch-bound on an MI355X: hundreds of small kernels per step.  Capturing forward
+ backward once and replaying it removes the per-launch host cost; the gradient all-reduce and the optimizer stay
outside the graph.  Static shapes only: every batch must have the shape of the batch the graph was captured with.  With
the hand-written training step (nn/train_engine.py) nothing on the step needs the host -- sequence lengths stay on the
device -- so ragged windows are captured like full ones; only the autograd fallback (configurations the engine does not
cover) still needs full-length windows, because packing sequences is a host round trip.
"""
import torch


class GraphedTrainStep(object):
    FIELDS = ('poses', 'shapes', 'trans', 'marker_pos_synth', 'marker_ori_synth', 'offset_t_augmented',
              'offset_r_augmented', 'joints_gt', 'seq_lengths')

    def __init__(self, net, optimizer, example_batch, warmup=3):
        import copy
        from em_pose_amd.nn.train_engine import LgdTrainEngine
        F = example_batch.seq_length
        engine = getattr(net, 'use_train_engine', True) and hasattr(net, 'N') and LgdTrainEngine.supported(net)
        if not engine and int(example_batch.seq_lengths.min()) != F:
            raise ValueError('a captured training step on the autograd fallback needs full-length windows')
        self.net, self.optimizer = net, optimizer
        self.static = copy.copy(example_batch)
        for k in self.FIELDS:
            v = getattr(example_batch, k, None)
            if torch.is_tensor(v):
                setattr(self.static, k, v.clone())
        # a gradient sink (helpers.distributed.GradientBuckets) still provides the static gradient storage, but must not
        # enqueue collectives while the step is traced: the caller runs `buckets.finish()` after every replay
        sink = getattr(net, '_grad_sink', None)
        if sink is not None:
            sink.hold = True
        was_full = getattr(net, 'full_windows', False)
        net.full_windows = True   # only while the step is traced: the replayed graph does not go through Python again
        # Warm-up and capture on the SAME side stream (allocator pools and any library state are per stream).
        # the warm-up passes must not count as training steps: BatchNorm running statistics are put back afterwards
        bn_state = {k: v.clone() for k, v in net.state_dict().items()
                    if 'running_' in k or k.endswith('num_batches_tracked')}
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                optimizer.zero_grad(set_to_none=True)
                out = net(self.static)
                net.backward(self.static, out, as_tensors=True)
            sd = net.state_dict()
            for k, v in bn_state.items():
                sd[k].copy_(v)
            side.synchronize()
            optimizer.zero_grad(set_to_none=True)
            del out
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=side):
            out = net(self.static)
            self.total, self.loss_vals = net.backward(self.static, out, as_tensors=True)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        net.full_windows = was_full
        self.sink = sink

    def load(self, batch):
        for k in self.FIELDS:
            v = getattr(batch, k, None)
            if torch.is_tensor(v):
                dst = getattr(self.static, k)
                if dst.shape != v.shape:
                    raise ValueError('batch field {} has shape {}, the graph was captured with {}'.format(
                        k, tuple(v.shape), tuple(dst.shape)))
                dst.copy_(v, non_blocking=True)

    def __call__(self, batch):
        """Replays forward + backward on `batch`; the parameter gradients are then in `.grad` (static tensors).
        :return: dict of loss values as device tensors (read them after the step, not inside it)"""
        self.load(batch)
        if self.sink is not None:
            self.sink.hold = True    # nothing of the replay goes through Python; finish() releases it
        self.graph.replay()
        # the replay ran no Python: BatchNorm running statistics moved without anybody noticing (and an optimizer on raw
        # pointers moves the weights the same way) -- invalidate what is cached from them (the folded inference handle)
        from em_pose_amd.nn import layers as _layers
        _layers.BN_STATS_GENERATION[0] += 1
        return self.loss_vals
