"""Dependency-free WebDataset / wids I/O for the precompute job (webdataset and wids are not installable here).

Formats (reference scripts/get_wids_input_json_para.py:36-51,
thinkdiff/tasks/image_text_process_data.py:70-118, thinkdiff/datasets/datasets/cc_sbu_dataset_mllama_vllm_process_wids.py:41):
  * shard = POSIX tar; a sample = consecutive members `<key>.<ext>` sharing `<key>`;
  * index = {"__kind__": "wids-shard-index-v1", "wids_version": 1, "name": ..., "shardlist": [{"url", "nsamples"}]};
  * precompute output sample = {__key__, jpg (JPEG bytes), json (utf-8), "<layer>.output_embed.pth",
    "<layer>.input_embed.pth"} with `.pth` = torch.save() bytes of a CPU tensor, shards named "%06d.tar",
    rolled over at maxsize bytes (5e8 in the reference) or maxcount samples.
"""
import io
import json
import os
import tarfile
import time
from typing import Any, Dict, List, Optional


def encode_value(ext: str, value: Any) -> bytes:
    """webdataset's default encoders for the extensions the job writes."""
    if isinstance(value, (bytes, bytearray)):
        return bytes(value)
    base = ext.split(".")[-1].lower()
    if base in ("jpg", "jpeg", "png"):
        buf = io.BytesIO()
        value.save(buf, format="JPEG" if base != "png" else "PNG")
        return buf.getvalue()
    if base == "json":
        return json.dumps(value).encode("utf-8")
    if base in ("txt", "text", "cls"):
        return str(value).encode("utf-8")
    if base in ("pth", "pt"):
        import torch
        buf = io.BytesIO()
        if hasattr(value, "save") and not torch.is_tensor(value):
            value.save(buf)                  # deferred serialisation (a view into a batch-wide tensor, cloned on the writer thread)
        else:
            torch.save(value, buf)
        return buf.getvalue()
    raise ValueError(f"no encoder for extension {ext!r} and value of type {type(value)}")


def decode_value(ext: str, data: bytes) -> Any:
    base = ext.split(".")[-1].lower()
    if base in ("jpg", "jpeg", "png"):
        from PIL import Image
        img = Image.open(io.BytesIO(data))
        img.load()
        return img
    if base == "json":
        return json.loads(data.decode("utf-8"))
    if base in ("txt", "text", "cls"):
        return data.decode("utf-8")
    if base in ("pth", "pt"):
        import torch
        return torch.load(io.BytesIO(data), map_location="cpu", weights_only=True)
    return data


def encode_sample(sample: Dict[str, Any]) -> Dict[str, Any]:
    """Every member of a sample as bytes (what TarWriter.write would produce); `__*` entries pass through.  Lets the JPEG /
    torch.save / json encoding run on worker threads while one thread appends to the tar in order."""
    return {ext: (value if ext.startswith("__") else encode_value(ext, value)) for ext, value in sample.items()}


class TarWriter:
    def __init__(self, path: str):
        self.path = path
        self._tar = tarfile.open(path, "w")
        self.size = 0

    def write(self, sample: Dict[str, Any]) -> int:
        key = sample["__key__"]
        total = 0
        now = time.time()
        for ext, value in sample.items():
            if ext.startswith("__"):
                continue
            data = encode_value(ext, value)
            info = tarfile.TarInfo(f"{key}.{ext}")
            info.size, info.mtime, info.mode, info.uname, info.gname = len(data), now, 0o444, "bigdata", "bigdata"
            self._tar.addfile(info, io.BytesIO(data))
            total += len(data)
        self.size += total
        return total

    def close(self):
        self._tar.close()


class ShardWriter:
    """`wds.ShardWriter(pattern, maxsize, maxcount, start_shard)` semantics (context manager, `.write(sample)`)."""

    def __init__(self, pattern: str, maxcount: int = 100000, maxsize: float = 3e9, start_shard: int = 0, verbose: int = 0):
        self.pattern, self.maxcount, self.maxsize, self.verbose = pattern, maxcount, maxsize, verbose
        self.shard = start_shard
        self.shards: List[Dict[str, Any]] = []
        self._w: Optional[TarWriter] = None
        self.count = self.size = self.total = 0
        self._next()

    def _next(self):
        self._finish()
        self.fname = self.pattern % self.shard
        self.shard += 1
        self._w = TarWriter(self.fname)
        self.count = self.size = 0

    def _finish(self):
        if self._w is not None:
            self._w.close()
            self.shards.append({"url": self.fname, "nsamples": self.count})
            self._w = None

    def write(self, sample: Dict[str, Any]):
        if self._w is None or self.count >= self.maxcount or self.size >= self.maxsize:
            self._next()
        self.size += self._w.write(sample)
        self.count += 1
        self.total += 1

    def close(self):
        self._finish()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def write_wids_index(path: str, shards: List[Dict[str, Any]], name: str = "dataset"):
    """reference scripts/get_wids_input_json_para.py:36-51"""
    with open(path, "w") as fh:
        json.dump({"__kind__": "wids-shard-index-v1", "wids_version": 1, "name": name,
                   "shardlist": [{"url": s["url"], "nsamples": int(s["nsamples"])} for s in shards]}, fh, indent=2)


def read_tar_samples(path: str, decode: bool = True) -> List[Dict[str, Any]]:
    """All samples of one shard, wids style: {'__key__', '.jpg', '.json', ...} (extension keys keep the dot)."""
    out: List[Dict[str, Any]] = []
    cur: Optional[Dict[str, Any]] = None
    with tarfile.open(path, "r") as tar:
        for m in tar:
            if not m.isfile():
                continue
            base = os.path.basename(m.name)
            dirn = os.path.dirname(m.name)
            stem, _, ext = base.partition(".")
            key = os.path.join(dirn, stem) if dirn else stem
            data = tar.extractfile(m).read()
            if cur is None or cur["__key__"] != key:
                cur = {"__key__": key, "__shard__": path}
                out.append(cur)
            cur["." + ext] = decode_value(ext, data) if decode else data
    return out


class ShardListDataset:
    """Map-style view of a wids index (`wids.ShardListDataset(json, keep=True, localname=identity)`)."""

    def __init__(self, index, shards: Optional[List[Dict[str, Any]]] = None, cache_shards: Optional[int] = None, chunksize: int = 1000):
        if shards is None:
            with open(index) as fh:
                desc = json.load(fh)
            assert desc.get("wids_version", 1) == 1, "unsupported wids index version"
            shards = desc["shardlist"]
            base = os.path.dirname(os.path.abspath(index))
            for s in shards:
                if not os.path.isabs(s["url"]) and not os.path.exists(s["url"]):
                    s["url"] = os.path.join(base, s["url"])
        self.shards = shards
        self.cum = [0]
        for s in shards:
            self.cum.append(self.cum[-1] + int(s["nsamples"]))
        self._cache: Dict[int, List[Dict[str, Any]]] = {}
        if cache_shards is None:
            # the chunked sampler shuffles inside windows of `chunksize` consecutive samples: keep every shard such a window can
            # touch (2 for the usual multi-thousand-sample shards; small shards would otherwise be re-read on every other access)
            smallest = max(1, min((int(s["nsamples"]) for s in shards), default=1))
            cache_shards = min(256, 2 + (chunksize + smallest - 1) // smallest)
        self._cache_shards = cache_shards

    def __len__(self):
        return self.cum[-1]

    def subset(self, rank: int, world: int) -> "ShardListDataset":
        """Rank's share of the SHARD list (`shardlist[rank::world]`): whole shards stay on one rank."""
        return ShardListDataset(None, shards=self.shards[rank::world], cache_shards=self._cache_shards)

    def _shard(self, si: int):
        if si not in self._cache:
            if len(self._cache) >= self._cache_shards:
                self._cache.pop(next(iter(self._cache)))
            samples = read_tar_samples(self.shards[si]["url"])
            assert len(samples) == int(self.shards[si]["nsamples"]), \
                f"{self.shards[si]['url']}: index says {self.shards[si]['nsamples']} samples, tar holds {len(samples)}"
            self._cache[si] = samples
        return self._cache[si]

    def __getitem__(self, i: int) -> Dict[str, Any]:
        if i < 0 or i >= len(self):
            raise IndexError(i)
        import bisect
        si = bisect.bisect_right(self.cum, i) - 1
        return self._shard(si)[i - self.cum[si]]


def chunked_order(n: int, chunksize: int = 1000, shuffle: bool = True, seed: int = 0) -> List[int]:
    """`wids.ChunkedSampler`: contiguous chunks (shard locality), shuffled between and within chunks."""
    import random
    rng = random.Random(seed)
    chunks = [list(range(s, min(n, s + chunksize))) for s in range(0, n, chunksize)]
    if shuffle:
        rng.shuffle(chunks)
        for c in chunks:
            rng.shuffle(c)
    return [i for c in chunks for i in c]
