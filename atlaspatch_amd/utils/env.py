"""Environment switches of this build (the reference has none: these select MI355X-side behaviour only)."""
from __future__ import annotations

import os


def env_flag(name: str, default: bool = False) -> bool:
    """``NAME`` unset or empty -> ``default``; ``0`` / ``false`` / ``no`` / ``off`` (any case) -> False; anything else -> True.
    One parser for every on / off variable, so ``NAME=0`` can never switch a feature ON."""
    raw = os.environ.get(name)
    if raw is None or raw.strip() == "":
        return default
    return raw.strip().lower() not in ("0", "false", "no", "off")
