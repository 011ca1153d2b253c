"""Where packaged data would live (the reference keeps its sample populations and genetic maps in
``breedgym/data``; this package ships none -- benchmarks synthesise their inputs, see synthetic.py)."""
from pathlib import Path

CODE_PATH = Path(__file__).resolve().parent.parent
DATA_PATH = CODE_PATH / "data"
PROJECT_PATH = CODE_PATH.parent
