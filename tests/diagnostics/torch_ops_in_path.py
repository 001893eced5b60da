"""which torch (aten) ops still run inside get_bn of the wav2vec2 tag: op name, count, Python call site"""
import sys, collections
sys.path.insert(0, ".")
import torch
import satools_amd
from satools_amd import synthetic
from torch.profiler import profile, ProfilerActivity

tag = sys.argv[1] if len(sys.argv) > 1 else "hifigan_bn_tdnnf_wav2vec2_vq_48_v1"
model = satools_amd.load_model("synthetic:" + tag)
model.to("cuda")
wav = synthetic.harm_batch(list(range(32))).to("cuda")
model.get_bn(wav)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    model.get_bn(wav)
    torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.name.startswith("aten::") and e.name not in ("aten::empty", "aten::view", "aten::as_strided", "aten::slice", "aten::select",
                                                       "aten::permute", "aten::reshape", "aten::empty_strided", "aten::_unsafe_view", "aten::unsqueeze", "aten::squeeze", "aten::expand", "aten::alias", "aten::detach", "aten::empty_like", "aten::stride", "aten::size", "aten::contiguous", "aten::to", "aten::is_nonzero", "aten::item", "aten::_local_scalar_dense"):
        site = next((s for s in (e.stack or []) if "sa-toolkit_amd" in s), "?")
        cnt[(e.name, site.strip()[-90:])] += 1
for (n, s), c in cnt.most_common(25):
    print(f"{c:5d}  {n:28s} {s}")
