"""`bs`-compatible command line for the hot path: `train`, `predict` and `segment` with the reference's
flags (/root/reference/bootstrapper/cli.py:51-92, predict.py:243-266, segment.py:166-241).
    python -m bootstrapper_amd.cli predict 02_pred.toml -s 01 -ng 8
    python -m bootstrapper_amd.cli segment 03_seg.toml -ws -p 'thresholds=[0.2,0.5]'
"""
import click

from .segment import load_toml, run_segmentation


@click.group()
def cli():
    """bootstrapper hot path on MI355X"""


@cli.command()
@click.argument("config_file", type=click.Path(exists=True, dir_okay=False))
@click.option("--setup-id", "-s", type=str, help="Setup ID(s) to run prediction for. 01, 02, etc.")
@click.option("--roi-offset", "-ro", type=str, help="Offset of ROI in world units (space separated integers)")
@click.option("--roi-shape", "-rs", type=str, help="Shape of ROI in world units (space separated integers)")
@click.option("--num-workers", "-nw", type=int, help="Number of workers")
@click.option("--num-gpus", "-ng", type=int, help="Number of GPUs to use")
@click.option("--precision", type=click.Choice(["bf16x3", "f32", "bf16"]), default="bf16x3", show_default=True,
              help="bf16x3: split bf16, within 1e-4 of the fp32 reference; f32: exact f32 MFMA; bf16: throughput mode (4e-3)")
def predict(config_file, setup_id, precision, **kwargs):
    """Run prediction for a setup or all setups in a prediction config file."""
    from .predict import run_prediction
    run_prediction(config_file, setup_id, precision=precision, **kwargs)


@cli.command()
@click.argument("config_file", type=click.Path(exists=True, dir_okay=False))
@click.option("--ws", "-ws", is_flag=True, help="Watershed segmentation")
@click.option("--mws", "-mws", is_flag=True, help="Mutex watershed segmentation")
@click.option("--cc", "-cc", is_flag=True, help="Connected componenents segmentation")
@click.option("--roi-offset", "-ro", type=str)
@click.option("--roi-shape", "-rs", type=str)
@click.option("--blockwise", "-b", is_flag=True, default=None)
@click.option("--num-workers", "-n", type=int)
@click.option("--block-shape", "-bs", type=str)
@click.option("--block-context", "-bc", type=str)
@click.option("--param", "-p", multiple=True, help="Method parameter override, e.g. -p 'thresholds=[0.2,0.3]'")
def segment(config_file, ws, mws, cc, **kwargs):
    """Segment affinities as specified in config_file."""
    config = load_toml(config_file)
    flagged = [m for m, on in (("ws", ws), ("mws", mws), ("cc", cc)) if on]
    if flagged:
        methods = flagged
    else:
        methods = [m for m in ("ws", "mws", "cc") if config.get(f"{m}_params")] or ["ws"]
    for method in methods:
        run_segmentation(config_file, method, **kwargs)


@cli.command()
@click.argument("config_file", type=click.Path(exists=True, dir_okay=False))
@click.option("--device", "-d", type=int, default=0, show_default=True)
def train(config_file, device):
    """Train the model of a setup directory as specified in config_file (fp32, one process per GPU; launch with
    torch.distributed.run for data-parallel training)."""
    import os
    import torch
    from .train import run_training
    if "RANK" in os.environ and not torch.distributed.is_initialized():
        torch.distributed.init_process_group("nccl")
        device = int(os.environ.get("LOCAL_RANK", device))
    run_training(config_file, device=device)


from .refine import refine as _refine  # noqa: E402

cli.add_command(_refine, "refine")

# aliases of the reference CLI (cli.py:38-44)
cli.add_command(train, "t")
cli.add_command(predict, "p")
cli.add_command(segment, "s")

if __name__ == "__main__":
    cli()
