"""Precompute job, host side: WebDataset shard writer/reader round trip, wids index, `.pth` members loadable by
torch.load, rank sharding of the shard list (no GPU: the model is a stub with the reference's output schema)."""
import io
import json
import os
import tarfile

import torch
from PIL import Image

from thinkdiff.datasets import wds_io
from thinkdiff.datasets.cc_sbu_process import CCSBUMllamaVllmProcessDatasetWids


def _make_input_shards(root, n_shards=3, per_shard=5):
    shards = []
    k = 0
    for s in range(n_shards):
        path = os.path.join(root, f"in-{s:05d}.tar")
        w = wds_io.TarWriter(path)
        for _ in range(per_shard):
            img = Image.new("RGB", (32 + k, 24), (k * 9 % 255, 10, 200))
            w.write({"__key__": f"sample{k:06d}", "jpg": img, "json": {"caption": f"caption {k}"}})
            k += 1
        w.close()
        shards.append({"url": path, "nsamples": per_shard})
    idx = os.path.join(root, "wids_shards.json")
    wds_io.write_wids_index(idx, shards, name="cc_sbu_test")
    return idx, k


class _StubModel:
    """Output schema of MllamaVllmGenerate_1.forward (reference mllama_vllm_generate_1.py:591-623)."""
    def __call__(self, samples):
        n = len(samples["images"])
        return {"generated_text": [f"text {i}" for i in range(n)],
                "generated_token": {"input_prompt": samples["answers"], "input_prompt_token_ids": [[1, 2, 3]] * n,
                                    "output_text": [f"text {i}" for i in range(n)], "output_token_ids": [(7, 8)] * n},
                "generated_embed": {"model.norm": {"output_embed": [torch.full((2, 8), float(i)).bfloat16() for i in range(n)],
                                                   "input_embed": [torch.full((3, 8), -float(i)).bfloat16() for i in range(n)]}}}


def test_wids_index_and_reader(tmp_path):
    idx, n = _make_input_shards(str(tmp_path))
    desc = json.load(open(idx))
    assert desc["__kind__"] == "wids-shard-index-v1" and desc["wids_version"] == 1 and len(desc["shardlist"]) == 3
    ds = wds_io.ShardListDataset(idx)
    assert len(ds) == n
    s = ds[7]
    assert s["__key__"] == "sample000007" and s[".json"]["caption"] == "caption 7" and s[".jpg"].size == (39, 24)
    # three members per input sample is what the reference's indexer assumes; here two (jpg, json) + none extra
    with tarfile.open(desc["shardlist"][0]["url"]) as t:
        assert [m.name for m in t][:2] == ["sample000000.jpg", "sample000000.json"]


def test_precompute_task_writes_reference_format(tmp_path):
    from thinkdiff.tasks.image_text_process_data import ImageTextProcessDataTask
    idx, n = _make_input_shards(str(tmp_path))
    ds = CCSBUMllamaVllmProcessDatasetWids(idx)
    order = wds_io.chunked_order(len(ds), chunksize=4, shuffle=True, seed=1)
    assert sorted(order) == list(range(n))
    loader = [ds.collater([ds[i] for i in order[s:s + 4]]) for s in range(0, n, 4)]
    out_dir = str(tmp_path / "out")
    stats = ImageTextProcessDataTask()._train_inner_loop(0, len(loader), _StubModel(), loader, output_shard_path=[out_dir, "%06d.tar", 5],
                                                           maxsize=4000)
    assert stats["samples"] == n and len(stats["shards"]) > 1                      # rolled over at maxsize
    assert os.path.basename(stats["shards"][0]["url"]) == "000005.tar"             # start_shard honoured
    seen = {}
    for sh in stats["shards"]:
        for s in wds_io.read_tar_samples(sh["url"]):
            seen[s["__key__"]] = s
    assert set(seen) == {f"sample{k:06d}" for k in range(n)}
    one = seen["sample000003"]
    assert set(one) >= {".jpg", ".json", ".model.norm.output_embed.pth", ".model.norm.input_embed.pth"}
    js = one[".json"]
    assert js["caption"] == "caption 3" and js["prompt"] in ds.instructions
    assert js["output_token_ids"] == [7, 8] and js["input_prompt_token_ids"] == [1, 2, 3] and "generated_text" in js
    emb = one[".model.norm.output_embed.pth"]
    assert emb.dtype == torch.bfloat16 and emb.shape == (2, 8) and emb.device.type == "cpu"
    # the raw member is a plain torch.save payload (what the reference's training dataset torch.loads)
    with tarfile.open(stats["shards"][0]["url"]) as t:
        m = [x for x in t if x.name.endswith("output_embed.pth")][0]
        assert torch.load(io.BytesIO(t.extractfile(m).read()), weights_only=True).shape == (2, 8)


def test_rank_partition_of_shards_is_disjoint_and_complete(tmp_path):
    idx, n = _make_input_shards(str(tmp_path), n_shards=5, per_shard=3)
    full = wds_io.ShardListDataset(idx)
    parts = [full.subset(r, 2) for r in range(2)]
    keys = [{p[i]["__key__"] for i in range(len(p))} for p in parts]
    assert keys[0].isdisjoint(keys[1]) and len(keys[0] | keys[1]) == n
    assert [len(p.shards) for p in parts] == [3, 2]


def test_pipelined_job_keeps_sample_order_compact_tensors_and_errors(tmp_path):
    """The job overlaps reading, the model and writing (prefetching loader, encoder threads, one tar thread): the output still holds
    the samples in loader order, every .pth member is the compact tensor of ITS sample (not the batch-wide storage it was a view
    of), small shards do not thrash the shard cache, and a failing model call propagates instead of hanging the writer."""
    from thinkdiff.runners.runner_process_data import _Loader
    from thinkdiff.tasks.image_text_process_data import ImageTextProcessDataTask
    idx, n = _make_input_shards(str(tmp_path), n_shards=6, per_shard=4)
    ds = CCSBUMllamaVllmProcessDatasetWids(idx)
    assert ds.inner_dataset._cache_shards >= 6                         # a sampler chunk (1000 samples) spans all six 4-sample shards
    reads = []
    orig = wds_io.read_tar_samples
    wds_io.read_tar_samples = lambda path, decode=True: (reads.append(path), orig(path, decode))[1]
    try:
        loader = _Loader(ds, batch_size=5, shuffle=True, seed=3)
        order = [s["filenames"] for s in loader]
    finally:
        wds_io.read_tar_samples = orig
    assert len(reads) == 6 and sum(len(b) for b in order) == n        # every shard read exactly once
    flat = [k for b in order for k in b]
    out_dir = str(tmp_path / "out")
    stats = ImageTextProcessDataTask()._train_inner_loop(0, len(loader), _StubModel(), _Loader(ds, batch_size=5, shuffle=True, seed=3),
                                                           output_shard_path=[out_dir, "%06d.tar", 0])
    written = [s for sh in stats["shards"] for s in wds_io.read_tar_samples(sh["url"], decode=False)]
    assert [s["__key__"] for s in written] == flat                      # tar order = loader order across the 5 batches
    pos_in_batch = {k: i for b in order for i, k in enumerate(b)}
    for s in written:
        raw = s[".model.norm.output_embed.pth"]
        assert len(raw) < 2000                                           # 2 x 8 bf16 values + pickle framing, not the whole batch
        t = torch.load(io.BytesIO(raw), weights_only=True)
        assert t.shape == (2, 8) and float(t[0, 0]) == float(pos_in_batch[s["__key__"]])      # _StubModel: sample i of its batch holds the value i

    class Boom(_StubModel):
        calls = 0

        def __call__(self, samples):
            Boom.calls += 1
            if Boom.calls == 2:
                raise RuntimeError("model failed on the second batch")
            return super().__call__(samples)
    import pytest
    with pytest.raises(RuntimeError, match="second batch"):
        ImageTextProcessDataTask()._train_inner_loop(0, len(loader), Boom(), _Loader(ds, batch_size=5, shuffle=True, seed=3),
                                                       output_shard_path=[str(tmp_path / "out2"), "%06d.tar", 0])
