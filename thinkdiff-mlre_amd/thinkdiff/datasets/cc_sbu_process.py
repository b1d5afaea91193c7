"""Input dataset of the embedding-precompute job (reference thinkdiff/datasets/datasets/
cc_sbu_dataset_mllama_vllm_process_wids.py:36-63): a wids shard list of {jpg, json} samples; the collater draws one
brief-description instruction per sample and returns {"images": [[PIL]], "answers": [str], "jsons": [dict],
"filenames": [key]}."""
import random
from typing import Any, Dict, List, Optional, Sequence

from .wds_io import ShardListDataset

# The instruction table of the job, carried as data in the reference's order (cc_sbu_dataset_mllama_vllm_process_wids.py:
# 11-27): `random.choice` under the same seed must pick the same instruction for the same sample, otherwise the generated
# text and the hidden-state shards are a different dataset.  11 brief-description + 5 diffusion-prompt instructions.
llava_brief_instructions = [
    "Describe the image concisely.",
    "Provide a brief description of the given image.",
    "Offer a succinct explanation of the picture presented.",
    "Summarize the visual content of the image.",
    "Give a short and clear explanation of the subsequent image.",
    "Share a concise interpretation of the image provided.",
    "Present a compact description of the photo's key features.",
    "Relay a brief, clear account of the picture shown.",
    "Render a clear and concise summary of the photo.",
    "Write a terse but informative summary of the picture.",
    "Create a compact narrative representing the image presented.",
    "Generate a prompt that can recreate the image in a 2D diffusion model.",
    "Provide a descriptive prompt to reproduce the given image using a diffusion model.",
    "Create a prompt suitable for a 2D diffusion model to generate the same image.",
    "Summarize the visual details as a prompt for a 2D diffusion model.",
    "Write a clear prompt to guide a 2D diffusion model in recreating the image.",
]
DEFAULT_INSTRUCTIONS = llava_brief_instructions      # `instructions=` overrides the table (tests, other jobs)


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
