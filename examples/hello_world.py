"""examples/hello_world.py of the reference: an A440 sine wave."""
from klang_amd.audio import Dac, Oscillator


def build():
    return Oscillator(frequency=440.) | Dac()


def main(argv=None, **kwargs):
    from examples import bounce
    bounce(build, 'hello_world.wav', 1.0, argv, **kwargs)


if __name__ == '__main__':
    main()
