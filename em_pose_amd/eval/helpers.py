"""
Evaluation driver (mirror of reference empose/eval/helpers.py:30-200 and scripts/evaluate_real.py:24-101).

  window_generator        256-frame chunks of one recording (reference helpers.py:30-48)
  load_model              config.json + model.pth of a released model id (reference helpers.py:131-164)
  partition_sequences     NEW: whole recordings -> ranks, longest-processing-time-first (chunks of one
                          recording are serially dependent through the LSTM state and must stay on one GPU)
  evaluate_sequences      the per-recording loop of evaluate_real.py (state carry, first chunk's shape for the whole
                          recording, metrics on all frames / per recording)
"""
import glob
import os
from collections import defaultdict

import numpy as np
import torch

from em_pose_amd.data.data import RealBatch
from em_pose_amd.eval.metrics import MetricsEngine
from em_pose_amd.helpers.configuration import CONSTANTS as C
from em_pose_amd.helpers.configuration import Configuration


def window_generator(batch, window_size):
    if window_size is None:
        yield batch
        return
    assert isinstance(batch, RealBatch)
    seq_len = batch.seq_length
    n_windows = seq_len // window_size + int(seq_len % window_size > 0)
    for i in range(n_windows):
        sf, ef = i * window_size, min((i + 1) * window_size, seq_len)
        lengths = torch.tensor([ef - sf], dtype=batch.seq_lengths.dtype, device=batch.seq_lengths.device)
        yield RealBatch(batch.ids, lengths, batch.poses[:, sf:ef], batch.shapes, batch.trans[:, sf:ef],
                        batch.marker_pos_real[:, sf:ef], batch.marker_ori_real[:, sf:ef], batch.marker_masks[:, sf:ef],
                        batch.offset_t, batch.offset_r)


def get_model_dir(experiment_dir, model_id):
    hits = glob.glob(os.path.join(experiment_dir, str(model_id) + '-*'))
    if len(hits) != 1:
        raise ValueError('expected exactly one model directory for id {} under {}, found {}'
                         .format(model_id, experiment_dir, len(hits)))
    return hits[0]


def load_model_weights(checkpoint_file, net, state_key='model_state_dict'):
    if not os.path.exists(checkpoint_file):
        raise ValueError('Could not find model checkpoint {}.'.format(checkpoint_file))
    ckpt = torch.load(checkpoint_file, map_location='cpu')[state_key]
    missing, unexpected = net.load_state_dict(ckpt, strict=False)
    bad = [k for k in list(missing) + list(unexpected) if not k.startswith('smpl.')]
    if bad:
        raise ValueError('checkpoint does not match the model: {}'.format(bad[:8]))


def get_all_offset_files():
    """{subject id: path} of the `*_offsets.npz` files of the real data set (reference helpers/utils.py:149-153)."""
    files = sorted(glob.glob(os.path.join(C.DATA_DIR_TEST, '*_offsets.npz')))
    return {os.path.split(f)[-1].split('_')[0]: f for f in files}


def sensor_vertex_ids():
    """The sensor sites of the asset tree: the `vertex_ids` of its offsets files (reference transforms.py:159, "the same
    for all offsets"), `C.VERTEX_IDS` when the tree has none.  On the licensed assets the two agree (the reference's
    network uses the constant, models.py:383, its preprocessing the files'); a tree around another mesh -- the 160-vertex
    one of tests/golden/eval_assets -- states its sites there."""
    ids = None
    for path in get_all_offset_files().values():
        here = [int(v) for v in np.load(path)['vertex_ids'].tolist()]
        if ids is not None and here != ids:
            raise ValueError('the offsets files under {} disagree about vertex_ids'.format(C.DATA_DIR_TEST))
        ids = here
    return list(C.VERTEX_IDS) if ids is None else ids


def load_model(model_id, device=None):
    """reference eval/helpers.py:148-164: config.json -> body model -> network -> model.pth (whose `smpl.bm.*` buffers
    replace what `model.npz` held, as `load_state_dict` does in the reference)."""
    from em_pose_amd.bodymodels.smpl import create_default_smpl_model
    from em_pose_amd.nn.models import create_model
    device = C.DEVICE if device is None else device
    model_dir = get_model_dir(C.EXPERIMENT_DIR, model_id)
    config = Configuration.from_json(os.path.join(model_dir, 'config.json'))
    smpl = create_default_smpl_model(device)
    net = create_model(config, smpl)
    if hasattr(net, 'vertex_ids'):
        net.vertex_ids = sensor_vertex_ids()
    load_model_weights(os.path.join(model_dir, 'model.pth'), net)
    return net.to(device).eval(), config, model_dir


def get_model_config(model_id):
    """(Configuration, model directory) of a released model (reference eval/helpers.py:140-145)."""
    model_dir = get_model_dir(C.EXPERIMENT_DIR, model_id)
    return Configuration.from_json(os.path.join(model_dir, 'config.json')), model_dir


def evaluate(data_loader, net, preprocess_fn, metrics_engine, window_size=None, device=None):
    """
    Losses and metrics over a hold-out set (reference eval/helpers.py:51-111): every batch is normalised as a whole,
    cut into temporal chunks (`window_size`, single-recording `RealBatch`es only, as in the reference), preprocessed,
    run through the model with the LSTM state carried over the chunks, and fed to the metrics engine with the shape of
    the first chunk.
    :return: dict of loss values averaged over the samples of the set.
    """
    device = C.DEVICE if device is None else device
    net.eval()
    keep = getattr(net, 'keep_history', True)
    net.keep_history = True            # the loss values are computed from the iteration histories
    agg, n_samples = defaultdict(float), 0
    metrics_engine.reset()
    try:
        with torch.no_grad():
            for b, abatch in enumerate(data_loader):
                abatch = preprocess_fn(abatch, mode='normalize_only')
                first_shape_hat, seq_vals, n_chunks, bs = None, defaultdict(float), 0, 0
                for i, achunk in enumerate(window_generator(abatch, window_size)):
                    chunk = preprocess_fn(achunk.to_gpu(device), mode='after_normalize', reset_rng=(i + b == 0))
                    out = net(chunk, is_new_sequence=(i == 0))
                    _, vals = net.backward(chunk, out)
                    for k, v in vals.items():
                        seq_vals[k] += v
                    pose_hat = out['pose_hat'] if out['pose_hat'] is not None else chunk.poses_body
                    if i == 0:
                        first_shape_hat = out['shape_hat'][:, 0] if out['shape_hat'] is not None else None
                    metrics_engine.compute(chunk.poses_body, chunk.shapes, pose_hat, first_shape_hat, chunk.seq_lengths,
                                           chunk.poses_root, out['root_ori_hat'], frame_mask=chunk.marker_masks)
                    n_chunks, bs = i + 1, chunk.batch_size
                for k, v in seq_vals.items():
                    agg[k] += v / n_chunks * bs
                n_samples += bs
    finally:
        net.keep_history = keep
    return {k: v / n_samples for k, v in agg.items()}


def compute_loss_and_metrics(data_loader, net, preprocess_fn, model_id, smpl_model=None, device=None):
    """reference eval/helpers.py:114-128"""
    if smpl_model is None:
        from em_pose_amd.bodymodels.smpl import create_default_smpl_model
        smpl_model = create_default_smpl_model(device)
    me = MetricsEngine(smpl_model)
    losses = evaluate(data_loader, net, preprocess_fn, me, device=device)
    print('[LOSS] loss: {:.6f}'.format(losses['total_loss']))
    metrics = me.get_metrics()
    print(me.to_pretty_string(metrics, model_id))
    return losses, metrics


def partition_sequences(lengths, world_size):
    """
    Longest-processing-time-first assignment of recordings to ranks.
    :return: list (per rank) of sorted recording indices; deterministic for equal lengths.
    """
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    load = [0] * world_size
    parts = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        parts[r].append(i)
        load[r] += int(lengths[i])
    return [sorted(p) for p in parts]


DEFER_FRAME_BUDGET = 65536   # evaluate_sequences: frames whose metrics may wait for one joint launch


def _check_async(dev):
    """After a synchronisation: a cooperative LSTM kernel that gave up on a poll poisoned its outputs with NaN and counted
    itself (include/empose_hip.h, empose_async_status) -- raise instead of averaging NaNs into the metrics table."""
    if torch.device(dev).type == 'cuda':
        from em_pose_amd import _lib
        _lib.check(_lib.lib().empose_async_status())


class _nothing(object):
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def evaluate_sequences(net, batches, smpl_model, device, window_size=256, log=None, host_times=None):
    """
    :param batches: iterable of single-recording `RealBatch`es (root already normalised), on the CPU.
    :return: (overall MetricsEngine, [(recording id, metrics dict)], frames processed)
    """
    from em_pose_amd.nn.models import IterativeErrorFeedback
    import time as _time
    _t = [_time.perf_counter()]

    def lap(key):   # (dev) host seconds per section of the loop
        if host_times is not None:
            now = _time.perf_counter()
            host_times[key] = host_times.get(key, 0.0) + now - _t[0]
            _t[0] = now
    # ONE engine collects the rows of every chunk of every recording (on the device, in call order); they are read back
    # once, after the last recording, and split by the per-recording counts of valid frames.  Reading them back per
    # recording (as the reference's two engines would) drains the two-stream pipeline 36 times: 3.3 ms each, 0.12 of the
    # 0.25 s of a pass.
    me_rows = MetricsEngine(smpl_model)
    ids, counts, frames = [], [], 0
    # ... and on the device path the metrics themselves are computed ONCE, over the frames of all chunks: a chunk only
    # leaves its tensors in `deferred` (the per-chunk forward-kinematics + metrics launches and the tensor plumbing around
    # them were 0.27 ms of host time per chunk, a quarter of a pass that is bound by the host).
    deferred, deferred_frames = [], 0
    defer = getattr(me_rows, 'angle_glob', False) and hasattr(smpl_model, 'fk_joints')

    def flush_deferred():
        # one forward-kinematics launch and one metrics launch over the frames of the deferred chunks, in chunk order; shapes
        # per frame (ground truth: the recording's; estimate: the recording's first chunk's, evaluate_real.py:63-68).  Called
        # at the end of the pass and whenever DEFER_FRAME_BUDGET frames have piled up (each deferred chunk pins its staging
        # buffer and outputs, ~1.8 KB per frame; the rows of a flush stay on the device until the pass ends, 65 doubles per frame)
        nonlocal deferred, deferred_frames
        if not deferred:
            return
        frames_of = lambda t: t.reshape(-1, t.shape[-1])
        cat = lambda k: torch.cat([frames_of(d[k]) for d in deferred]).unsqueeze(0)
        per_frame = lambda k, alt: torch.cat([(d[k] if d[k] is not None else d[alt]).reshape(1, -1)
                                              .expand(d[0].shape[1], -1) for d in deferred]).unsqueeze(0)
        me_rows.compute(cat(0), per_frame(1, 1), cat(2), per_frame(3, 1), None, cat(4), cat(5),
                        valid=torch.cat([d[6].reshape(-1) for d in deferred]).unsqueeze(0))
        deferred, deferred_frames = [], 0
    is_lgd = isinstance(net, IterativeErrorFeedback)
    ws = window_size if is_lgd else None
    # Chunks of a recording depend on each other only through the LSTM state, so chunk c + 1's packing + LSTM (current
    # stream) runs beside chunk c's refinement iterations and metrics (a side stream, IterativeErrorFeedback.iter_stream).
    # Same results as one chunk after the other.
    dev = torch.device(device)
    pipelined = is_lgd and dev.type == 'cuda' and not net.training
    side = None
    if pipelined:
        # one side stream per network, kept: a fresh stream per call may land on the hardware queue of the current
        # stream (measured: the first pass pipelined, later passes with their new streams did not)
        side = getattr(net, '_side_stream', None)
        if side is None or side.device != dev:
            side = net._side_stream = torch.cuda.Stream(device=dev)
        net.iter_stream = side
    try:
        for batch in batches:
            if log:
                log('Evaluate {} ({} frames)'.format(batch.ids[0], int(batch.seq_lengths[0])))
            first_shape_hat = None
            ids.append(batch.ids[0])
            counts.append(0)
            # (Staging the whole recording on the device once and slicing views was measured slower: recordings have 36
            # different lengths, and every new size costs the caching allocator a ~50 ms hipMalloc/hipFree round.)
            for c, chunk in enumerate(window_generator(batch, ws)):
                frames += int(chunk.seq_lengths.sum())   # read while the lengths are still on the host
                # the frames that count, while lengths and masks are still on the host (no device kernels for it)
                valid = MetricsEngine.valid_frames(chunk.seq_lengths, chunk.batch_size, chunk.seq_length,
                                                   chunk.marker_masks)
                counts[-1] += int(valid.sum())
                lap('cut_chunk')
                chunk = chunk.to_gpu(device)
                lap('to_gpu')
                out = net(chunk, is_new_sequence=(c == 0))
                lap('forward_enqueue')
                if side is not None and net.outputs_ready is None:
                    # the forward did not go through the two-stream path (autograd forward of a `differentiable` net with
                    # grad enabled): its outputs are on the current stream, the metrics below are on the side stream
                    side.wait_stream(torch.cuda.current_stream(dev))
                if defer and side is not None and out['pose_hat'].is_cuda:
                    if c == 0:  # the first chunk's shape is used for the whole recording (evaluate_real.py:63-68)
                        first_shape_hat = out['shape_hat'][:, 0] if out['shape_hat'] is not None else None
                    deferred.append((chunk.poses_body, chunk.shapes, out['pose_hat'], first_shape_hat, chunk.poses_root,
                                     out['root_ori_hat'], valid))
                    deferred_frames += chunk.seq_length
                    if deferred_frames >= DEFER_FRAME_BUDGET:   # bound what is kept alive: row order is unchanged
                        torch.cuda.current_stream(dev).wait_stream(side)
                        flush_deferred()
                    lap('metrics_enqueue')
                    continue
                with torch.cuda.stream(side) if side is not None else _nothing():
                    if side is not None:   # the chunk lives in memory of the current stream's pool
                        for t in (chunk.poses, chunk.shapes, chunk.seq_lengths):
                            t.record_stream(side)
                    if c == 0:  # the first chunk's shape is used for the whole recording (evaluate_real.py:63-68)
                        first_shape_hat = out['shape_hat'][:, 0] if out['shape_hat'] is not None else None
                    # the reference feeds the same chunk to two engines (evaluate_real.py:70-81); compute once per chunk
                    me_rows.compute(chunk.poses_body, chunk.shapes, out['pose_hat'], first_shape_hat, chunk.seq_lengths,
                                   chunk.poses_root, out['root_ori_hat'], frame_mask=chunk.marker_masks, valid=valid)
                lap('metrics_enqueue')
        if side is not None:
            torch.cuda.current_stream(dev).wait_stream(side)   # every chunk's outputs / rows are complete
        flush_deferred()
        st = me_rows.state()        # (reads the rows back: the device is in sync afterwards)
        _check_async(dev)
        lap('wait_for_device_and_rows')
        me_all, per_sequence, at = MetricsEngine(smpl_model), [], 0
        for sid, m in zip(ids, counts):
            me_ind = MetricsEngine(smpl_model)
            me_ind.merge({key: v[at:at + m] for key, v in st.items()})
            at += m
            me_all.merge(me_ind.state())
            per_sequence.append((sid, me_ind.get_metrics()))
        lap('per_recording_metrics')
    finally:
        if pipelined:
            net.iter_stream = None
    return me_all, per_sequence, frames


def evaluate_sequences_batched(net, batches, smpl_model, device, window_size=256, host_times=None):
    """
    Same results as `evaluate_sequences`, but chunk c of ALL recordings runs as one ragged batch (rows = recordings
    that still have frames, `seq_lengths` = frames left in this chunk, LSTM state carried per row).  Windows are
    independent of each other in the model, so this only changes how much work one launch carries:
    36 recordings need 14 launches of the loop instead of 211.

    The metric rows of a chunk are scattered on the device to where they belong in ONE block in recording order; a
    recording's rows go to the host (pinned memory, side stream) with the chunk it ends in, and its table row is worked
    out while the device runs the chunks after it -- the host part of the metrics (a third of a pass when it ran after the
    last chunk) is hidden behind the device, which is the bound of this driver (the LSTM's dependent steps).

    `host_times` (a dict, optional): seconds of host time per section of the pass (dev: where a slow pass spends it).
    """
    import time as _time
    _t = [_time.perf_counter()]

    def lap(key):
        if host_times is not None:
            now = _time.perf_counter()
            host_times[key] = host_times.get(key, 0.0) + now - _t[0]
            _t[0] = now
    from em_pose_amd.nn.models import IterativeErrorFeedback
    assert isinstance(net, IterativeErrorFeedback)
    if len(batches) == 0:      # (a rank of a multi-GPU run that got no recording)
        return MetricsEngine(smpl_model), [], 0
    # rows shorter than the chunk are padded: average the shape over their valid frames only, which is what the
    # unpadded one-recording chunk of the sequential driver averages over
    was_valid_only = net.shape_avg_valid_only
    net.shape_avg_valid_only = True
    try:
        import numpy as np
        from torch.nn.utils.rnn import pad_sequence
        n = len(batches)
        lengths = [int(b.seq_lengths[0]) for b in batches]
        # Longest recording first: the rows that still have frames in chunk c are then a PREFIX of the batch, so a chunk is
        # a slice [:k, sf:sf + f] of one padded block per field (no per-row cutting, padding and concatenating: that was
        # 47 of the 82 ms of a pass, all of it host time -- the device waits for the host in this driver).
        order = sorted(range(n), key=lambda i: (-lengths[i], i))
        sl = [lengths[i] for i in order]
        n_chunks = (sl[0] + window_size - 1) // window_size
        dev = torch.device(device)
        on_gpu = dev.type == 'cuda'
        # (a recording a loader has pinned -- RealBatch.pin_memory -- goes up as ONE block, without staging and without
        # blocking the host: 36 copies per pass instead of 288)
        on_dev = [batches[i].device_fields(device) for i in order]
        fields = ('poses', 'trans', 'marker_pos_real', 'marker_ori_real', 'marker_masks')
        packed = {k: pad_sequence([d[k][0] for d in on_dev], batch_first=True) for k in fields}
        whole = {k: torch.cat([d[k] for d in on_dev]) for k in ('shapes', 'offset_t', 'offset_r')}
        del on_dev
        lap('upload_and_pack')
        # the frames that count -- inside the recording and every sensor present -- from the host copies of the masks
        # (numpy: a torch CPU op on a 36 x 256 x 12 block wakes the whole intra-op thread pool)
        valid_all = np.zeros((n, sl[0]), dtype=bool)
        for j, i in enumerate(order):
            valid_all[j, :sl[j]] = (batches[i].marker_masks[0].numpy() != 0).all(axis=-1)
        # As in `evaluate_sequences`: chunk c + 1's packing and LSTM (current stream) beside chunk c's refinement
        # iterations and metrics (side stream); nothing in the loop waits for the device.
        side = None
        if on_gpu and not net.training:
            side = getattr(net, '_side_stream', None)
            if side is None or side.device != dev:
                side = net._side_stream = torch.cuda.Stream(device=dev)
            net.iter_stream = side
        lap('valid_frames')
        me_chunks = MetricsEngine(smpl_model)
        placed = on_gpu and me_chunks.angle_glob and hasattr(smpl_model, 'fk_joints')   # (the device path of `compute`)
        # Where every frame's row goes: recording i owns rows base[i] .. base[i + 1] of one block, its valid frames in
        # frame order (what the sequential driver accumulates, recording after recording); frames that do not count go to
        # a spare row past the end.
        per_row = valid_all.sum(axis=1)
        base = np.zeros(n + 1, dtype=np.int64)
        base[1:][order] = per_row
        base = np.concatenate([[0], np.cumsum(base[1:])])
        total = int(base[-1])
        if placed:
            dest_all = np.where(valid_all, np.cumsum(valid_all, axis=1) - 1 + base[:-1][order][:, None], total)
            dest_host = torch.from_numpy(np.ascontiguousarray(dest_all, dtype=np.int64)).pin_memory()
            dest_dev = dest_host.to(device, non_blocking=True)
            rows_dev = torch.empty(total + 1, 65, dtype=torch.float64, device=dev)
            rows_host = torch.empty(max(total, 1), 65, dtype=torch.float64, pin_memory=True)
            rows_np = rows_host.numpy()[:total]   # (the view keeps the pinned block alive; torch caches it afterwards)
        counts = []                             # not placed: per chunk, the valid frames of each of its rows
        arrived = []                            # placed: (event, recordings whose rows it brings), in chunk order
        results = [None] * n                    # per recording: its table row

        def table_row(i):
            me = MetricsEngine(smpl_model)
            r = rows_np[base[i]:base[i + 1]]
            me.merge({'eucl': r[:, :22], 'eucl_pa': r[:, 22:44], 'angle': r[:, 44:]})
            results[i] = (batches[i].ids[0], me.get_metrics())

        def take_arrived(wait):
            while arrived and (wait or arrived[0][0].query()):
                ev, recs = arrived.pop(0)
                ev.synchronize()
                for i in recs:
                    table_row(i)
        # the lengths of every chunk's rows in ONE pinned block and one copy that does not block (a pageable copy would make
        # the host wait for the stream; a pinned block per chunk asks the host allocator 14 times a pass, and a request it
        # cannot serve from its cache waits for the device)
        lens_all = np.clip(np.asarray(sl, dtype=np.int64)[None, :] - window_size * np.arange(n_chunks)[:, None], 0,
                           window_size).astype(np.int32)
        lens_all_dev = torch.from_numpy(lens_all)
        if on_gpu:
            lens_all_dev = lens_all_dev.pin_memory().to(device, non_blocking=True)
        state, first_shape, frames = None, None, 0
        lap('row_plan')
        for c in range(n_chunks):
            sf = c * window_size
            k = sum(1 for L in sl if L > sf)
            k_next = sum(1 for L in sl if L > sf + window_size)
            lens = [min(window_size, L - sf) for L in sl[:k]]
            f = lens[0]
            cut = lambda name: packed[name][:k, sf:sf + f]
            lens_dev = lens_all_dev[c, :k]
            chunk = RealBatch([batches[i].ids[0] for i in order[:k]], lens_dev, cut('poses'),
                              whole['shapes'][:k], cut('trans'), cut('marker_pos_real'), cut('marker_ori_real'),
                              cut('marker_masks'), whole['offset_t'][:k], whole['offset_r'][:k]).to_gpu(device)
            valid_np = valid_all[:k, sf:sf + f]
            if net.rnn_init and c > 0:   # the rows that go on are the first k of the previous chunk's
                net.rnn.final_state = (state[0][:, :k].contiguous(), state[1][:, :k].contiguous())
            lap('chunk_prep')
            out = net(chunk, is_new_sequence=(c == 0))
            lap('forward_enqueue')
            if net.rnn_init:
                state = net.rnn.final_state
            if side is not None and net.outputs_ready is None:   # (see evaluate_sequences)
                side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side) if side is not None else _nothing():
                if side is not None:   # the chunk lives in memory of the current stream's pool
                    for t in (chunk.poses, chunk.shapes, chunk.seq_lengths):
                        t.record_stream(side)
                if c == 0:   # the first chunk's shape estimate stands for the whole recording (evaluate_real.py:63-68)
                    first_shape = out['shape_hat'][:, 0].contiguous()
                me_chunks.compute(chunk.poses_body, chunk.shapes, out['pose_hat'], first_shape[:k], chunk.seq_lengths,
                                  chunk.poses_root, out['root_ori_hat'], frame_mask=chunk.marker_masks,
                                  valid=torch.from_numpy(np.ascontiguousarray(valid_np)))
                if placed:
                    (rows, _, _), = me_chunks.take_device_rows()
                    if side is not None:
                        dest_dev.record_stream(side)
                        rows_dev.record_stream(side)
                    rows_dev.index_copy_(0, dest_dev[:k, sf:sf + f].reshape(-1), rows)
                    ended = [order[j] for j in range(k_next, k)]   # the recordings whose last chunk this was
                    for i in ended:
                        if base[i + 1] > base[i]:
                            rows_host[base[i]:base[i + 1]].copy_(rows_dev[base[i]:base[i + 1]], non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record()
                    arrived.append((ev, ended))
            if not placed:
                counts.append(valid_np.sum(axis=1).tolist())
            frames += sum(lens)
            lap('metrics_enqueue')
            take_arrived(wait=False)
            lap('per_recording_metrics')
        if placed:
            take_arrived(wait=True)     # (the last event is the end of the side stream's work)
            lap('wait_for_device')
        if side is not None:
            torch.cuda.current_stream(dev).wait_stream(side)
        if on_gpu:
            torch.cuda.synchronize(dev)
            _check_async(dev)
        lap('wait_for_device')
        if placed:
            for i in range(n):          # (a recording without a single frame never ended in a chunk: an empty table row)
                if results[i] is None:
                    table_row(i)
            me_all = MetricsEngine(smpl_model)
            me_all.merge({'eucl': rows_np[:, :22], 'eucl_pa': rows_np[:, 22:44], 'angle': rows_np[:, 44:]})
            return me_all, results, frames
        st = me_chunks.state()    # one device-side gather of the valid rows, one copy to (cached) pinned host memory
        engines = [MetricsEngine(smpl_model) for _ in range(n)]
        at = 0
        for cnt in counts:
            for j, m in enumerate(cnt):
                if m:
                    engines[order[j]].merge({key: v[at:at + m] for key, v in st.items()})
                at += m
    finally:
        net.shape_avg_valid_only = was_valid_only
        net.iter_stream = None
    lap('merge_rows')
    me_all = MetricsEngine(smpl_model)
    per_sequence = []
    for i in range(n):  # recording order, exactly as the sequential driver accumulates
        me_all.merge(engines[i].state())
        per_sequence.append((batches[i].ids[0], engines[i].get_metrics()))
    lap('per_recording_metrics')
    return me_all, per_sequence, frames
