"""Input dataset of the embedding-precompute job (reference thinkdiff/datasets/datasets/
cc_sbu_dataset_mllama_vllm_process_wids.py:36-63): a wids shard list of {jpg, json} samples; the collater draws one
brief-description instruction per sample and returns {"images": [[PIL]], "answers": [str], "jsons": [dict],
"filenames": [key]}."""
import random
from typing import Any, Dict, List, Optional, Sequence

from .wds_io import ShardListDataset

# The reference ships its own list of 16 brief-description instructions (cc_sbu_dataset_mllama_vllm_process_wids.py:
# 13-33); any list can be supplied through `instructions=`.  These defaults are ours.
DEFAULT_INSTRUCTIONS = [
    "Describe the image concisely.",
    "Give a brief description of the picture.",
    "Summarize what the image shows in one or two sentences.",
    "Write a short caption that captures the content of the photo.",
]


class CCSBUMllamaVllmProcessDatasetWids:
    def __init__(self, location: str, instructions: Optional[Sequence[str]] = None, rank: int = 0, world: int = 1):
        ds = ShardListDataset(location)
        self.inner_dataset = ds.subset(rank, world) if world > 1 else ds
        self.instructions = list(instructions or DEFAULT_INSTRUCTIONS)

    def __len__(self):
        return len(self.inner_dataset)

    def __getitem__(self, i):
        return self.inner_dataset[i]

    def collater(self, samples: List[Dict[str, Any]]) -> Dict[str, Any]:
        images, answers, jsons, filenames = [], [], [], []
        for s in samples:
            images.append([s[".jpg"].convert("RGB")])
            prompt = random.choice(self.instructions)
            answers.append(prompt)
            js = dict(s[".json"])
            js["prompt"] = prompt
            jsons.append(js)
            filenames.append(s["__key__"])
        return {"images": images, "answers": answers, "jsons": jsons, "filenames": filenames}
