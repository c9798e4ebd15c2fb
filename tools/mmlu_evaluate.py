"""script/3-mmlu-evaluate.py restated for local paths (no Lightning, no downloads):

    python tools/mmlu_evaluate.py --ckpt .data/opt-1.3b.ckpt --spt_ckpt .data/opt-1.3b-spt.ckpt \\
        --data_root ~/Public/Datasets/text --tokenizer /path/to/saved/tokenizer

``--data_root`` holds ``mmlu/{dev,test,val,auxiliary_train}/*.csv``; ``--tokenizer`` is a directory a
``transformers`` tokenizer was saved to, or ``bytes`` (a byte-level stand-in: it runs the pipeline,
its numbers mean nothing for a model trained on another vocabulary).  Prints one JSON line."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'spt-proto_amd'))
import torch  # noqa: E402
from naive_gpt import loaders, utils  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ckpt', default='.data/opt-125m.ckpt', help='base model checkpoint {config, state_dict}')
    ap.add_argument('--spt_ckpt', default=None, help='tuned adapters (state_dict over the upgraded model)')
    ap.add_argument('--data_root', default=os.path.join(os.path.expanduser('~'), 'Public/Datasets/text'))
    ap.add_argument('--tokenizer', default='bytes')
    ap.add_argument('--seq_length', type=int, default=512)
    ap.add_argument('--batch_size', type=int, default=1)
    ap.add_argument('--test_batches', type=int, default=64)
    ap.add_argument('--n_shots', type=int, default=5)
    ap.add_argument('--d_lora', type=int, default=16)
    ap.add_argument('--device', default='cuda' if torch.cuda.is_available() else 'cpu')
    args = ap.parse_args()
    dm = loaders.MMLUDataModule(root=args.data_root, n_shots=args.n_shots, batch_size=args.batch_size,
                                num_workers=0, tokenizer=args.tokenizer, seq_length=args.seq_length + 1)
    model = utils.load_spt_model(args.ckpt, args.spt_ckpt, d_lora=args.d_lora, device=args.device)
    result = utils.evaluate_mmlu(model, dm.test_dataloader(), n_batches=args.test_batches, device=args.device)
    print(json.dumps(dict(result, ckpt=args.ckpt, spt_ckpt=args.spt_ckpt, tokenizer=args.tokenizer)))


if __name__ == '__main__':
    main()
