"""Drop-in for the reference's main.py (main.py:1-8): same two objects, same calls."""
from dvi_ekf_amd import Config, Simulator

config = Config("config.yaml")
sim = Simulator(config)

if __name__ == "__main__":
    sim.run_once()
    sim.plot(compact=True)
