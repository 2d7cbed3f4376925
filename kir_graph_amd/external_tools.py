"""
External binaries (samtools / hisat2) are outside the accelerated path
(``graphkir/external_tools.py`` is process plumbing).
Only the local engine is provided; ``--engine`` is accepted for CLI parity.
"""
from __future__ import annotations

from .utils import runShell

_engine = "local"


def setEngine(engine: str) -> None:
    global _engine
    if engine != "local":
        raise NotImplementedError(
            f"--engine {engine}: container engines are not part of this build; install the tool locally")
    _engine = engine


def runTool(name: str, command: list[str], capture_output: bool = False, cwd: str | None = None):
    return runShell(command, capture_output=capture_output, cwd=cwd)
