"""World-size-2 rehearsal of the data-parallel path on CPU (gloo): parameter / BatchNorm-buffer broadcast from
rank 0 and the bucketed gradient all-reduce driven by the backward stage hook.  The kernels themselves need a
GPU; here the per-stage gradient slices are filled by hand exactly where kasf_backward would have written them."""
import ctypes as C
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker_bf16(rank, world, port, overlap, q):
    """grad_dtype="bf16": the bucket travels as bf16 and comes back widened; fp32 master gradients elsewhere untouched."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import kasportsformer_amd as K
        from kasportsformer_amd import _lib
        torch.manual_seed(7)
        m = K.KASportsFormer(n_layers=2, num_heads=8, n_frames=27)
        opt = K.FusedAdamW(m)
        dp = K.DataParallel(m, overlap=overlap, optimizer=opt, grad_dtype="bf16")
        lib = _lib.load()
        gen = torch.Generator().manual_seed(1000 + rank)
        g = torch.zeros(m.n_flat)
        mine = torch.randn(m.n_flat, generator=gen) * 1e-3
        b, e = C.c_int64(), C.c_int64()
        for st in range(lib.kasf_backward_stages(m._layout)):
            _lib.check(lib.kasf_stage_grad_range(m._layout, st, C.byref(b), C.byref(e)))
            if e.value > b.value:
                g[b.value:e.value] = mine[b.value:e.value]
                if m.grad_stage_hook is not None:
                    m.grad_stage_hook(st, g[b.value:e.value])
        m.flat_grad = g
        dp.finish_gradients()
        # what the wire format allows: each rank's contribution rounded to bf16, summed (in bf16 by the collective), widened
        parts = [(torch.randn(m.n_flat, generator=torch.Generator().manual_seed(1000 + r)) * 1e-3).bfloat16() for r in range(world)]
        expect = sum(p.float() for p in parts)
        err = (g[:m.n_live] - expect[:m.n_live]).abs().max() / expect[:m.n_live].abs().max()
        assert float(err) < 2 ** -7, float(err)             # one more bf16 rounding of the sum
        assert torch.count_nonzero(g[m.n_live:]) == 0 and g.dtype == torch.float32
        assert opt.grad_scale == 1.0 / world
        # the staging buffers are keyed by bucket ORDER: later steps -- whose flat gradient is a fresh allocation at another address (another batch size, an
        # evaluation pass in between) -- reuse them instead of adding a set per address
        n_wire, ptrs = len(dp._wire), sorted(v.data_ptr() for v in dp._wire.values())
        keep_alive = []
        for step in range(3):
            g2 = torch.zeros(m.n_flat)
            keep_alive.append(g2)                          # keeps the allocator from handing the same address out again
            assert g2.data_ptr() != g.data_ptr()
            for st in range(lib.kasf_backward_stages(m._layout)):
                _lib.check(lib.kasf_stage_grad_range(m._layout, st, C.byref(b), C.byref(e)))
                if e.value > b.value:
                    g2[b.value:e.value] = mine[b.value:e.value]
                    if m.grad_stage_hook is not None:
                        m.grad_stage_hook(st, g2[b.value:e.value])
            m.flat_grad = g2
            dp.finish_gradients()
            assert len(dp._wire) == n_wire and sorted(v.data_ptr() for v in dp._wire.values()) == ptrs, (step, len(dp._wire), n_wire)
            assert torch.equal(g2[:m.n_live], g[:m.n_live])
        q.put((rank, "ok"))
    except Exception as ex:  # pragma: no cover
        q.put((rank, repr(ex)))
    finally:
        dist.destroy_process_group()


def _worker(rank, world, port, overlap, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import kasportsformer_amd as K
        from kasportsformer_amd import _lib
        torch.manual_seed(100 + rank)                       # different init per rank: broadcast must fix it
        m = K.KASportsFormer(n_layers=2, num_heads=8, n_frames=27)
        m._flat_buffers.uniform_(0, 1)
        m._nbt.fill_(rank + 5)
        dp = K.DataParallel(m, overlap=overlap)
        ref = [torch.zeros_like(m._flat), torch.zeros_like(m._flat_buffers), torch.zeros_like(m._nbt)]
        for t, src in zip(ref, (m._flat, m._flat_buffers, m._nbt)):
            t.copy_(src)
            dist.broadcast(t, src=0)
        assert all(torch.equal(a, b) for a, b in zip(ref, (m._flat, m._flat_buffers, m._nbt))), "ranks differ after sync_from_rank0"
        assert m.layers_with_bone[1].att_spatial.mixer.qkv.weight.data_ptr() >= m._flat.data_ptr()   # still views of the flat array
        # emulate kasf_backward stage by stage: rank r contributes (r+1) * pattern to every live gradient
        lib = _lib.load()
        g = torch.zeros(m.n_flat)
        pattern = torch.arange(m.n_flat, dtype=torch.float32) % 97
        stages = lib.kasf_backward_stages(m._layout)
        b, e = C.c_int64(), C.c_int64()
        covered = torch.zeros(m.n_flat, dtype=torch.bool)
        for st in range(stages):
            _lib.check(lib.kasf_stage_grad_range(m._layout, st, C.byref(b), C.byref(e)))
            if e.value > b.value:
                g[b.value:e.value] = (rank + 1) * pattern[b.value:e.value]
                covered[b.value:e.value] = True
                if m.grad_stage_hook is not None:
                    m.grad_stage_hook(st, g[b.value:e.value])
        m.flat_grad = g
        if world > 1:
            try:
                dp.finish_gradients()                       # nobody would divide the summed gradient by world_size
                raise AssertionError("finish_gradients without an optimizer must raise at world_size > 1")
            except RuntimeError:
                pass
        opt = K.FusedAdamW(m)
        dp.finish_gradients(opt)
        assert opt.grad_scale == 1.0 / world
        assert covered[:m.n_live].all() and not covered[m.n_live:].any()      # dead norm1_limb tail is never reduced
        expect = sum(r + 1 for r in range(world)) * pattern
        assert torch.equal(g[:m.n_live], expect[:m.n_live])
        assert torch.count_nonzero(g[m.n_live:]) == 0
        # INTEGRATION path A under data parallelism: a stock torch.optim optimizer has no grad_scale, so finish_gradients turns the all-reduced SUM into
        # the MEAN in place -- in the flat array the per-parameter .grad tensors are views of
        g2 = torch.zeros(m.n_flat)
        for st in range(stages):
            _lib.check(lib.kasf_stage_grad_range(m._layout, st, C.byref(b), C.byref(e)))
            if e.value > b.value:
                g2[b.value:e.value] = (rank + 1) * pattern[b.value:e.value]
                if m.grad_stage_hook is not None:
                    m.grad_stage_hook(st, g2[b.value:e.value])
        m.flat_grad = g2
        for prm, off, n, shape in m._live:                  # what _launch_backward leaves with attach_param_grads=True
            prm.grad = g2[off:off + n].view(shape)
        stock = torch.optim.SGD(m.parameters(), lr=1.0)
        before = m._flat.clone()
        dp.finish_gradients(stock)
        assert torch.allclose(g2[:m.n_live], expect[:m.n_live] / world, rtol=1e-6, atol=0)
        prm, off, n, shape = m._live[3]
        assert torch.equal(prm.grad.flatten(), g2[off:off + n])
        stock.step()
        for prm, off, n, shape in m._live[::37]:            # (the flat layout has alignment gaps no parameter covers: compare where parameters are)
            assert torch.allclose(m._flat[off:off + n], before[off:off + n] - expect[off:off + n] / world, rtol=1e-5, atol=1e-5)
        m.attach_param_grads = False
        try:
            dp.attach_optimizer(torch.optim.SGD(m.parameters(), lr=1.0))
            raise AssertionError("a stock optimizer with attach_param_grads=False would step on nothing")
        except RuntimeError:
            m.attach_param_grads = True
        # training leaves every rank with ITS OWN BatchNorm running statistics; evaluation and checkpoints must see rank 0's on every rank
        # (nn.DataParallel keeps replica 0's buffers, train_and_evaluate_sp.py:262-264)
        from kasportsformer_amd.evaluate import _broadcast_buffers
        for how in ("evaluate", "checkpoint"):
            m._flat_buffers.fill_(float(rank + 1))
            m._nbt.fill_(rank + 11)
            if how == "evaluate":
                _broadcast_buffers(m, None)                 # what evaluate_one_epoch(distributed=True) does first
            else:
                import tempfile
                path = os.path.join(tempfile.gettempdir(), f"kasf_dp_gloo_{port}.pth")
                K.checkpoint_save(path, 0, 5e-4, None, m, 1.0, "id", data_parallel=dp)
                dist.barrier()
                sd = K.strip_module_prefix(torch.load(path, map_location="cpu", weights_only=True)["model"])
                key = "layers_with_bone.0.graph_spatial.mixer.batch_norm.running_mean"
                assert float(sd[key][0]) == 1.0 and int(sd[key.replace("running_mean", "num_batches_tracked")]) == 11
                dist.barrier()
                if rank == 0:
                    os.remove(path)
            assert float(m._flat_buffers.min()) == 1.0 and float(m._flat_buffers.max()) == 1.0 and int(m._nbt[0]) == 11, (how, rank)
        q.put((rank, "ok"))
    except Exception as ex:  # pragma: no cover
        q.put((rank, repr(ex)))
    finally:
        dist.destroy_process_group()


def _run(overlap, worker=None):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=worker or _worker, args=(r, 2, port, overlap, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(r[1] == "ok" for r in res), res


def test_data_parallel_bucketed_allreduce_gloo():
    _run(overlap=True)


def test_data_parallel_single_allreduce_gloo():
    _run(overlap=False)


def test_data_parallel_bf16_gradient_allreduce_gloo():
    _run(overlap=True, worker=_worker_bf16)
    _run(overlap=False, worker=_worker_bf16)
