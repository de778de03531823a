// test stand-in, see ../README.md
#pragma once
