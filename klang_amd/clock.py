"""Global play-time clock shared by blocks (klang/clock.py:4-23)."""


class ClockMixin:
    currentTime = 0.

    @classmethod
    def set_current_time(cls, time):
        cls.currentTime = time

    @classmethod
    def clock(cls):
        return cls.currentTime
