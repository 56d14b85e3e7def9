"""Feature-list parsing and the skip-existing probe (reference: utils/features.py:10-71)."""
from __future__ import annotations

from pathlib import Path
from typing import Sequence

import click

from .h5 import h5


def parse_feature_list(raw: str, *, choices: list[str]) -> list[str]:
    names = [tok.strip().lower() for tok in raw.replace(",", " ").split() if tok.strip()]
    if not names:
        raise click.BadParameter("At least one feature extractor name is required.")
    unknown = [n for n in names if n not in choices]
    if unknown:
        raise click.BadParameter(f"Unknown extractor(s): {', '.join(unknown)}. Available: {', '.join(choices)}")
    ordered, repeated = [], []
    for n in names:
        (repeated if n in ordered else ordered).append(n)
    if repeated:
        raise click.BadParameter(f"Duplicate extractor(s) specified: {', '.join(sorted(set(repeated)))}. "
                                 "Provide each extractor at most once.")
    return ordered


def get_existing_features(h5_path, *, expected_total: int | None = None) -> set[str]:
    """Names (lower-cased) of complete feature datasets; unreadable/missing file -> empty set."""
    try:
        with h5.File(Path(h5_path), "r") as f:
            if "features" not in f:
                return set()
            found = set()
            for name, ds in f["features"].items():
                if expected_total is not None:
                    try:
                        if int(ds.shape[0]) != int(expected_total):
                            continue
                    except Exception:  # noqa: BLE001
                        continue
                found.add(str(name).lower())
            return found
    except Exception:  # noqa: BLE001
        return set()


def missing_features(h5_path, required: Sequence[str], *, expected_total: int | None = None) -> list[str]:
    have = get_existing_features(h5_path, expected_total=expected_total)
    return [name for name in (r.lower() for r in required) if name not in have]
