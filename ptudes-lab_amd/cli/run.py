"""`ptudes`-compatible entry point (reference src/ptudes/cli/run.py): `python -m ptudes_lab_amd.cli.run ekf-bench ...`"""
import click

from .ekf_bench import ptudes_ekf_bench
from .stat import ptudes_stat


@click.group(name="ptudes")
def ptudes_cli() -> None:
    """P(oint) (e)Tudes: MI355X-native pose path of ptudes-lab."""


ptudes_cli.add_command(ptudes_ekf_bench)
ptudes_cli.add_command(ptudes_stat)


def main():
    ptudes_cli()


if __name__ == "__main__":
    main()
