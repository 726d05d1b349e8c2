"""`indextts` command line (same flags and exit codes as the reference's indextts/cli.py:10-58)."""
import os
import sys


def main(argv=None):
    import argparse
    parser = argparse.ArgumentParser(description="IndexTTS command line (MI355X build)")
    parser.add_argument("text", type=str, help="text to synthesise")
    parser.add_argument("-v", "--voice", type=str, required=True, help="reference (prompt) audio file, wav")
    parser.add_argument("-o", "--output_path", type=str, default="gen.wav", help="output wav path")
    parser.add_argument("-c", "--config", type=str, default="checkpoints/config.yaml", help="config file")
    parser.add_argument("--model_dir", type=str, default="checkpoints", help="model directory")
    parser.add_argument("--fp16", action="store_true", default=True, help="half-precision GPT weights when available")
    parser.add_argument("-f", "--force", action="store_true", default=False, help="overwrite the output file")
    parser.add_argument("-d", "--device", type=str, default=None, help="device (cuda:N); CPU/MPS are not supported")
    args = parser.parse_args(argv)

    def fail(msg):
        print(f"[error] {msg}")
        parser.print_help()
        sys.exit(1)

    if len(args.text.strip()) == 0:
        fail("text is empty")
    if not os.path.exists(args.voice):
        fail(f"reference audio {args.voice} does not exist")
    if not os.path.exists(args.config):
        fail(f"config {args.config} does not exist")
    if os.path.exists(args.output_path):
        if not args.force:
            fail(f"output file {args.output_path} exists; use --force to overwrite")
        os.remove(args.output_path)
    try:
        import torch
    except ImportError:
        print("[error] PyTorch is not installed")
        sys.exit(1)
    if args.device is None:
        if not torch.cuda.is_available():
            print("[error] no GPU visible: this build needs an AMD GPU (no CPU path)")
            sys.exit(1)
        args.device = "cuda:0"
    from indextts.infer import IndexTTS
    tts = IndexTTS(cfg_path=args.config, model_dir=args.model_dir, is_fp16=args.fp16, device=args.device)
    tts.infer(audio_prompt=args.voice, text=args.text.strip(), output_path=args.output_path)


if __name__ == "__main__":
    main()
