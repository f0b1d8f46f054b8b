import sys, os, torch, warnings
from collections import OrderedDict
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
warnings.simplefilter("ignore")
from flamo_amd.graph import GraphedStep
from flamo_amd import ops
from flamo_amd.processor import dsp, system
dev = torch.device('cuda:0')
if os.environ.get('MEMHIST'):
    torch.cuda.memory._record_memory_history(max_entries=200000)
mode = sys.argv[1]
for f in sys.argv[2:]:
    mod, name = f.split('.')
    setattr(ops if mod == 'ops' else system, name, False)
torch.manual_seed(1)
nfft, N = 192000, 16
kw = dict(nfft=nfft, alias_decay_db=30.0, device=dev, dtype=torch.float32)
x = torch.randn(1, nfft, 1, device=dev)
c = torch.randn(1, nfft, 1, device=dev)
ig = dsp.Gain(size=(N, 1), requires_grad=True, **kw)
og = dsp.Gain(size=(1, N), requires_grad=True, **kw)
mods = OrderedDict(input_gain=ig)
if mode in ("delay", "rec", "rec_noatt"):
    dl = dsp.parallelDelay(size=(N,), max_len=3000, isint=True, **kw)
if mode == "delay":
    mods["d"] = dl
if mode in ("rec", "rec_noatt"):
    mix = dsp.Matrix(size=(N, N), matrix_type="orthogonal", requires_grad=True, **kw)
    if mode == "rec":
        att = dsp.parallelGEQ(size=(N,), requires_grad=True, **kw)
        att.map = dsp.db_of_sigmoid
        with torch.no_grad(): att.param.copy_(torch.randn_like(att.param) * 0.3 + 2.0)
        fb = system.Series(OrderedDict(mixing_matrix=mix, attenuation=att))
    else:
        fb = mix
        with torch.no_grad(): ig.param.mul_(0.1)
    mods["feedback_loop"] = system.Recursion(fF=dl, fB=fb)
mods["output_gain"] = og
out_layer = dsp.iFFT(nfft) if mode == "plain_ifft" else dsp.iFFTAntiAlias(nfft, alias_decay_db=30.0, device=dev)
model = system.Shell(system.Series(mods), dsp.FFT(nfft), out_layer)
params = [p for p in model.parameters() if p.requires_grad]
keep = {}
def fn(xx):
    s0 = torch.cuda.current_stream()
    y = model(xx)
    s1 = torch.cuda.current_stream()
    cap = torch.cuda.is_current_stream_capturing()
    if cap:
        keep["y"] = y
    yc = y * c
    if cap:
        keep["yc"] = yc
    r = yc.sum()
    if cap:
        keep["alts"] = {"f64": yc.sum(dtype=torch.float64), "rows": yc.reshape(375, 512).sum(dim=1), "mean": yc.mean(),
                        "sq": (yc * yc).sum(), "abs_max": yc.abs().max(), "half": yc.reshape(-1)[:96000].sum()}
    if cap:
        print("  capture: stream before/after model:", s0.cuda_stream, s1.cuda_stream, "capturing after:", cap,
              "ptrs y/yc/out: %x %x %x" % (y.data_ptr(), yc.data_ptr(), r.data_ptr()), "c %x x %x" % (c.data_ptr(), xx.data_ptr()))
    return r
gs = GraphedStep(fn, (x,), params, warmup=2)
out0 = gs.replay().clone()
torch.cuda.synchronize()
y0 = keep["y"].clone()
print("y layout", keep["y"].shape, keep["y"].stride(), keep["y"].storage_offset(), "storage elems", keep["y"].untyped_storage().nbytes() // 4)
pre = torch.zeros(4096, device=dev)
vals = []
for i in range(3):
    out = gs.replay(); torch.cuda.synchronize()
    which = os.environ.get("BETWEEN", "alloc")
    if which == "alloc":
        j = [torch.full((n,), 5.0, device=dev) for n in (1, 8, 512, 4096) for _ in range(8)]; del j
    elif which == "fill_only":
        pre.fill_(float(i))
    elif which == "alloc_nofill":
        j = [torch.empty((n,), device=dev) for n in (1, 8, 512, 4096) for _ in range(8)]; del j
    elif which == "one_small":
        j = torch.full((1,), float(os.environ.get("JUNKVAL", "5.0")), device=dev)
        ptr = j.data_ptr()
        for seg in torch.cuda.memory_snapshot():
            if seg["address"] <= ptr < seg["address"] + seg["total_size"]:
                print("  junk block at %x in segment %x size %d pool_id %s stream %s" % (ptr, seg["address"], seg["total_size"], seg.get("segment_pool_id"), seg.get("stream")))
        for nm, t in (("out", gs.static_out), ("y", keep["y"])):
            p2 = t.data_ptr()
            for seg in torch.cuda.memory_snapshot():
                if seg["address"] <= p2 < seg["address"] + seg["total_size"]:
                    print("  %s at %x in segment %x size %d pool_id %s" % (nm, p2, seg["address"], seg["total_size"], seg.get("segment_pool_id")))
        if os.environ.get('MEMHIST') and i == 0:
            snap = torch.cuda.memory._snapshot()
            evs = [e for tr in snap.get("device_traces", []) for e in tr if e.get("addr", 0) <= ptr < e.get("addr", 0) + max(e.get("size", 0), 1)]
            print("  events at junk address:", len(evs))
            for e in evs[-10:]:
                fr = [f"{f['filename'].split('/')[-1]}:{f['line']}:{f['name']}" for f in e.get("frames", []) if "flamo" in f["filename"] or "tools" in f["filename"] or "graphs.py" in f["filename"]][:8]
                print("   ", e.get("action"), e.get("size"), "stream", e.get("stream"), fr)
        del j
    elif which == "one_big":
        j = torch.full((1 << 22,), 5.0, device=dev); del j
    vals.append(out.clone())
    ref = {"f64": keep["yc"].double().sum(), "rows": keep["yc"].reshape(375, 512).sum(dim=1), "mean": keep["yc"].mean(),
           "sq": (keep["yc"] * keep["yc"]).sum(), "abs_max": keep["yc"].abs().max(), "half": keep["yc"].reshape(-1)[:96000].sum()}
    print("  alts ok:", {k: bool(torch.allclose(v.double(), ref[k].double(), rtol=1e-4)) for k, v in keep["alts"].items()})
    print("  replay", i, "yc == y*c eager:", torch.equal(keep["yc"], keep["y"] * c), "eager sum of graph's yc:", keep["yc"].sum().item(), "y same as first:", torch.equal(keep["y"], y0), "eager (y*c).sum():", (keep["y"] * c).sum().item(), "max|y-y0|", (keep["y"] - y0).abs().max().item())
print(mode, [v.item() for v in vals], "first", out0.item(), "| sum(x*c)", (x * c).sum().item(), "sum(c)", c.sum().item(), "5*sum(c)", 5 * c.sum().item(), "sum(y0*c)[:n]", [(y0 * c).reshape(-1)[:n].sum().item() for n in (512, 1024, 4096, 65536)])
