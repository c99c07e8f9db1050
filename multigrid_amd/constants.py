"""Index tables of the reference's enums (multigrid/core/constants.py:32-113, core/actions.py:4-15)."""

TYPE_NAMES = ("unseen", "empty", "wall", "floor", "door", "key", "ball", "box", "goal", "lava", "agent")
COLOR_NAMES = ("red", "green", "blue", "purple", "yellow", "grey")
STATE_NAMES = ("open", "closed", "locked", "pickedup", "available", "alive", "dead")
ACTION_NAMES = ("left", "right", "forward", "pickup", "drop", "toggle", "done")
DIRECTION_NAMES = ("right", "down", "left", "up")

UNSEEN, EMPTY, WALL, FLOOR, DOOR, KEY, BALL, BOX, GOAL, LAVA, AGENT = range(11)
RED, GREEN, BLUE, PURPLE, YELLOW, GREY = range(6)
OPEN, CLOSED, LOCKED, PICKEDUP, AVAILABLE, ALIVE, DEAD = range(7)
LEFT, RIGHT, FORWARD, PICKUP, DROP, TOGGLE, DONE = range(7)

KIND_EMPTY, KIND_REDBLUEDOORS, KIND_BLOCKEDUNLOCKPICKUP, KIND_LOCKEDHALLWAY, KIND_COMPETITIVE = range(5)
