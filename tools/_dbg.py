import sys, numpy as np, torch
sys.path.insert(0, ".")
from kube_scheduler_rs_reference_amd import FIT, PICK_SAMPLED, SEL, Evaluator, _lib, synth
from oracle import capi
c = synth.make_cluster(3000, 2500, n_keys=8, n_taints=0, seed=0x6A); pc = c.pod_columns()
flags = FIT | SEL | PICK_SAMPLED
o_feas, _, o_bind = capi.eval_encoded(c.avail_cpu, c.avail_mem, c.node_labels, None, pc["req_cpu_milli"], pc["req_mem_bytes"], pc["sel_val_ids"], None, pc["samples"], flags)
dev = torch.device("cuda", 0)
t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).view(dt)).to(dev)
for how in (2, 4, 8, 2, 4, 8, 5, 2, 4, 8):
    with Evaluator(0) as ev:
        ev.set_nodes(**c.node_columns())
        d = (t(pc["req_cpu_milli"], np.int64), t(pc["req_mem_bytes"], np.int64), t(pc["sel_val_ids"], np.int32), None, t(pc["samples"], np.int32))
        mask = ev.alloc_mask(c.P, how=how)
        bind = torch.full((c.P,), -9, dtype=torch.int32, device=dev)
        for kernel in ("fused", "direct", "fused"):
            ev.set_kernel(kernel)
            mask.fill_(-1)
            ev.eval_device(*d, flags, out_feasible=mask, out_binding=bind)
            torch.cuda.synchronize()
            got = mask.contiguous().cpu().numpy().view(np.uint64)
            bad = np.argwhere((got != o_feas).any(axis=1)).ravel()
            got2 = mask._base.cpu().numpy().view(np.uint64)[:c.P, :ev.W]  # the whole buffer through a DMA copy (no shader read)
            bad2 = np.argwhere((got2 != o_feas).any(axis=1)).ravel()
            if bad.size or bad2.size:
                print(f"   shader-read wrong rows {bad.size}, DMA-read wrong rows {bad2.size}; first wrong rows {bad[:8]} / {bad2[:8]}; row {bad[0] if bad.size else -1}: got {got[bad[0]][:4] if bad.size else None} want {o_feas[bad[0]][:4] if bad.size else None}")
                import time
                for trial in range(3):
                    torch.cuda.synchronize(); time.sleep(0.05)
                    g3 = mask.clone().cpu().numpy().view(np.uint64)
                    print(f"   later shader read {trial}: wrong rows {(g3 != o_feas).any(axis=1).sum()}")
                junk = torch.empty(64 << 20, dtype=torch.uint8, device=dev); junk.fill_(1); torch.cuda.synchronize(); del junk; torch.cuda.empty_cache()
                g4 = mask.clone().cpu().numpy().view(np.uint64)
                print(f"   after an unrelated 64 MiB allocation + free (page-table activity): wrong rows {(g4 != o_feas).any(axis=1).sum()}")
                # write side: does a torch kernel's write land where the DMA engine reads?
                mask.fill_(7); torch.cuda.synchronize()
                g5 = mask._base.cpu().numpy().view(np.uint64)[:c.P, :ev.W]
                print(f"   torch fill_(7) seen by the DMA read: rows != 7: {(g5 != 7).any(axis=1).sum()} of {c.P}")
            print(f"how {how} {kernel}: ptr {mask.data_ptr():#x} base {mask._base.data_ptr():#x} rows wrong {bad.size}" + (f" first {bad[0]} last {bad[-1]}; untouched rows {(got == np.uint64(0xFFFFFFFFFFFFFFFF)).all(axis=1).sum()}" if bad.size else ""), flush=True)
        del mask
