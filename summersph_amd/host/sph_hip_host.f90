! sph_hip_host.f90 -- thin Fortran host over the MI355X SPH core.
!
! Keeps the reference's public face so that a user of SUMMER_SPH.f90 can switch:
!   types   particle, sink            (same component names as SUMMER_SPH.f90:14-37)
!   ingest  read_data_from_file       (same text format and sink convention, :594-716)
!   loop    simulate(bodies, sinks)   (same step sequence, dt control, save cadence, :863-930)
!   saves   make_save                 (same columns, :719-738)
! Everything numerically heavy happens on the GPU through sph_hip_binding; this module only
! moves data across the boundary and keeps the book-keeping of the time loop.
!
! The device does everything simulate() does per step: density, EOS, Barnes-Hut gas self-gravity,
! sink gravity, SPH forces, kicks, drift, dt control, sink accretion and the boundary cull (the
! particle count may shrink; bodies is re-sized when that happens).
! Deliberate differences to the reference (see DESIGN.md "host"):
!   * make_save writes one record per line with an explicit format (the reference's
!     list-directed output wraps lines under flang) and replaces an existing file.
!   * simulate takes optional arguments (end time, step limit, quiet) for testing.
module sph_hip_host
  use, intrinsic :: iso_c_binding
  use sph_hip_binding
  use sph_hip_textio
  implicit none
  private
  public :: dp, particle, sink, read_data_from_file, simulate, make_save
  public :: bounding_size, smoothing

  integer, parameter :: dp = kind(1.0d0)
  real(dp), parameter :: smoothing = 2.5_dp, bounding_size = 1500.0_dp

  type :: particle
    integer :: number
    real(dp) :: mass
    real(dp) :: density
    real(dp) :: internal_energy
    real(dp) :: pressure
    real(dp) :: sound_speed
    real(dp) :: internal_energy_rate
    real(dp) :: alpha
    real(dp) :: alpha_rate
    real(dp), dimension(3) :: position
    real(dp), dimension(3) :: velocity
    real(dp), dimension(3) :: acceleration
  end type particle

  type :: sink
    real(dp) :: mass
    real(dp) :: radius
    real(dp), dimension(3) :: spin
    real(dp), dimension(3) :: position
    real(dp), dimension(3) :: velocity
    real(dp), dimension(3) :: acceleration
  end type sink

contains

  ! ------------------------------------------------------------------------------------------
  ! Ingest.  One header line, then records whose first eight values are
  ! x y z vx vy vz u m (further columns ignored).  u == 0 marks a sink (radius 3.5); gas
  ! particles get alpha = 0 and number = their order among the gas rows.  Without any sink
  ! row a single massless sink at the origin is created, as the reference does.
  ! ------------------------------------------------------------------------------------------
  subroutine read_data_from_file(filename, bodies, sinks)
    character(len=*), intent(in) :: filename
    type(particle), allocatable, intent(inout) :: bodies(:)
    type(sink), allocatable, intent(inout) :: sinks(:)
    real(dp), allocatable :: rec(:, :), grown(:, :)
    character(len=1024) :: line
    integer :: unit_no, ios, nrec, ngas, nsink, k, ig, is, pos, nread, nlines, buffered
    logical :: wrapped, at_line_end, more
    real(dp) :: v(8)

    open(newunit=unit_no, file=filename, status='old', action='read', iostat=ios)
    if (ios /= 0) then
      write(*, *) 'Error opening file: ', trim(filename)
      return
    end if
    read(unit_no, '(A)', iostat=ios) line          ! header
    allocate(rec(8, 4096))
    nrec = 0
    pos = len(line) + 1                            ! nothing buffered yet
    buffered = 0
    do
      ! A record is a run of values that may continue over line breaks, as for the reference's list-directed read
      ! ([F]:647).  One record per line: the first eight values count, further columns are skipped.  A record wrapped
      ! over several lines (a save written with list-directed output under flang: 80 columns per line) carries
      ! 9 values for a gas particle (..., alpha) and 8 for a sink.  Framing rule, independent of the values read: every
      ! record starts on a new line and a wrapped record's first line holds as many values as fit (at least two at 80
      ! columns), so when the 8th value ends a line, a following line that holds a SINGLE value is the 9th value of
      ! this record and never the start of the next one.  A file with one value per line cannot be framed that way and
      ! is rejected instead of being guessed at.
      call next_values(unit_no, line, pos, v, nread, nlines, ios)
      if (nread == 0) exit                         ! end of file between records
      if (ios /= 0) then
        write(*, *) 'Error reading line ', nrec + 1
        exit
      end if
      wrapped = nlines + buffered >= 2
      if (nlines + buffered >= 8) then
        write(*, *) 'Error: ', trim(filename), ' holds one value per line near record ', nrec + 1, &
                    ': records cannot be framed (8 or 9 values?); write one record per line'
        nrec = 0
        exit
      end if
      at_line_end = tokens_left(line, pos) == 0
      pos = len(line) + 1                          ! what is left of the line belongs to this record
      buffered = 0
      if (wrapped .and. at_line_end) then
        call load_line(unit_no, line, pos, more)
        if (more) then
          if (tokens_left(line, 1) == 1) then
            pos = len(line) + 1                    ! the wrapped alpha
          else
            buffered = 1                           ! first line of the next record
          end if
        end if
      end if
      if (nrec == size(rec, 2)) then
        allocate(grown(8, 2 * nrec))
        grown(:, 1:nrec) = rec
        call move_alloc(grown, rec)
      end if
      nrec = nrec + 1
      rec(:, nrec) = v
    end do
    close(unit_no)
    if (nrec == 0) then
      write(*, *) 'No data found in file: ', trim(filename)
      return
    end if

    nsink = count(rec(7, 1:nrec) == 0.0_dp)
    ngas = nrec - nsink
    if (allocated(bodies)) deallocate(bodies)
    if (allocated(sinks)) deallocate(sinks)
    allocate(bodies(ngas), sinks(max(nsink, 1)))
    ig = 0
    is = 0
    do k = 1, nrec
      if (rec(7, k) /= 0.0_dp) then
        ig = ig + 1
        bodies(ig)%position = rec(1:3, k)
        bodies(ig)%velocity = rec(4:6, k)
        bodies(ig)%internal_energy = rec(7, k)
        bodies(ig)%mass = rec(8, k)
        bodies(ig)%alpha = 0.0_dp
        bodies(ig)%alpha_rate = 0.0_dp
        bodies(ig)%number = ig
        bodies(ig)%density = 0.0_dp
        bodies(ig)%pressure = 0.0_dp
        bodies(ig)%sound_speed = 0.0_dp
        bodies(ig)%internal_energy_rate = 0.0_dp
        bodies(ig)%acceleration = 0.0_dp
      else
        is = is + 1
        sinks(is)%position = rec(1:3, k)
        sinks(is)%velocity = rec(4:6, k)
        sinks(is)%mass = rec(8, k)
        sinks(is)%radius = 3.5_dp
        sinks(is)%spin = 0.0_dp
        sinks(is)%acceleration = 0.0_dp
      end if
    end do
    if (nsink == 0) then
      sinks(1)%position = 0.0_dp
      sinks(1)%velocity = 0.0_dp
      sinks(1)%acceleration = 0.0_dp
      sinks(1)%spin = 0.0_dp
      sinks(1)%mass = 0.0_dp
      sinks(1)%radius = 0.0_dp
    end if
    write(*, *) 'Successfully read ', size(bodies), ' bodies and ', size(sinks), ' sinks from ', trim(filename), '.'
  end subroutine read_data_from_file

  ! ------------------------------------------------------------------------------------------
  ! Snapshot: header, then x y z vx vy vz u m alpha per gas particle and
  ! x y z vx vy vz 0 m per sink -- the reference's columns, one record per line, 17 digits.
  ! ------------------------------------------------------------------------------------------
  subroutine make_save(bodies, sinks, number, filename)
    type(particle), intent(in) :: bodies(:)
    type(sink), intent(in) :: sinks(:)
    integer, intent(in) :: number
    character(len=*), intent(in), optional :: filename
    character(len=256) :: savename
    integer :: io, i

    if (present(filename)) then
      savename = filename
    else
      write(savename, '(A,I0,A)') 'save', number, '.txt'
    end if
    open(newunit=io, file=trim(savename), status='replace', action='write')
    write(io, '(A)') 'x  y  z  vx  vy  vz  energy  mass  alpha'
    do i = 1, size(bodies)
      write(io, '(9(1X,ES25.17E3))') bodies(i)%position, bodies(i)%velocity, bodies(i)%internal_energy, &
        bodies(i)%mass, bodies(i)%alpha
    end do
    do i = 1, size(sinks)
      write(io, '(8(1X,ES25.17E3))') sinks(i)%position, sinks(i)%velocity, 0.0_dp, sinks(i)%mass
    end do
    close(io)
  end subroutine make_save

  ! ------------------------------------------------------------------------------------------
  ! device <-> host hand-over
  ! ------------------------------------------------------------------------------------------
  subroutine check(ctx, status, what)
    type(c_ptr), intent(in) :: ctx
    integer(c_int), intent(in) :: status
    character(len=*), intent(in) :: what
    if (status /= SPH_OK) then
      write(*, *) 'summersph: ', what, ' failed: ', c_message(sph_strerror(status)), ' -- ', c_message(sph_last_error(ctx))
      error stop 1
    end if
  end subroutine check

  subroutine push_state(ctx, bodies, sinks)
    type(c_ptr), intent(in) :: ctx
    type(particle), intent(in) :: bodies(:)
    type(sink), intent(in) :: sinks(:)
    real(c_double), allocatable :: a(:, :), s(:, :)
    integer :: n, ns
    n = size(bodies)
    ns = size(sinks)
    allocate(a(max(n, 1), 9), s(ns, 7))
    a(1:n, 1) = bodies%position(1); a(1:n, 2) = bodies%position(2); a(1:n, 3) = bodies%position(3)
    a(1:n, 4) = bodies%velocity(1); a(1:n, 5) = bodies%velocity(2); a(1:n, 6) = bodies%velocity(3)
    a(1:n, 7) = bodies%internal_energy; a(1:n, 8) = bodies%mass; a(1:n, 9) = bodies%alpha
    call check(ctx, sph_upload(ctx, int(n, c_int64_t), a(:, 1), a(:, 2), a(:, 3), a(:, 4), a(:, 5), a(:, 6), &
                               a(:, 7), a(:, 8), a(:, 9)), 'sph_upload')
    s(:, 1) = sinks%position(1); s(:, 2) = sinks%position(2); s(:, 3) = sinks%position(3)
    s(:, 4) = sinks%velocity(1); s(:, 5) = sinks%velocity(2); s(:, 6) = sinks%velocity(3)
    s(:, 7) = sinks%mass
    call check(ctx, sph_set_sinks(ctx, int(ns, c_int32_t), s(:, 1), s(:, 2), s(:, 3), s(:, 4), s(:, 5), s(:, 6), s(:, 7)), &
               'sph_set_sinks')
    call check(ctx, sph_set_sink_radii(ctx, int(ns, c_int32_t), sinks%radius), 'sph_set_sink_radii')
  end subroutine push_state

  subroutine pull_state(ctx, bodies, sinks, with_derived)
    type(c_ptr), intent(in) :: ctx
    type(particle), intent(inout) :: bodies(:)
    type(sink), intent(inout) :: sinks(:)
    logical, intent(in) :: with_derived
    real(c_double), allocatable :: a(:, :), s(:, :)
    integer :: n, ns, i
    integer(c_int64_t) :: n64
    n = size(bodies)
    ns = size(sinks)
    n64 = int(n, c_int64_t)
    allocate(a(max(n, 1), 9), s(ns, 10))
    call check(ctx, sph_download_state(ctx, n64, a(:, 1), a(:, 2), a(:, 3), a(:, 4), a(:, 5), a(:, 6), a(:, 7), a(:, 8), a(:, 9)), &
               'sph_download_state')
    do i = 1, n
      bodies(i)%position = a(i, 1:3)
      bodies(i)%velocity = a(i, 4:6)
      bodies(i)%internal_energy = a(i, 7)
      bodies(i)%mass = a(i, 8)
      bodies(i)%alpha = a(i, 9)
    end do
    if (with_derived .and. n > 0) then
      call check(ctx, sph_download_field(ctx, SPH_F_RHO, a(:, 1), n64), 'download rho'); bodies%density = a(1:n, 1)
      call check(ctx, sph_download_field(ctx, SPH_F_P, a(:, 1), n64), 'download P'); bodies%pressure = a(1:n, 1)
      call check(ctx, sph_download_field(ctx, SPH_F_C, a(:, 1), n64), 'download c'); bodies%sound_speed = a(1:n, 1)
      call check(ctx, sph_download_field(ctx, SPH_F_AX, a(:, 1), n64), 'download ax'); bodies%acceleration(1) = a(1:n, 1)
      call check(ctx, sph_download_field(ctx, SPH_F_AY, a(:, 1), n64), 'download ay'); bodies%acceleration(2) = a(1:n, 1)
      call check(ctx, sph_download_field(ctx, SPH_F_AZ, a(:, 1), n64), 'download az'); bodies%acceleration(3) = a(1:n, 1)
      call check(ctx, sph_download_field(ctx, SPH_F_DU, a(:, 1), n64), 'download du'); bodies%internal_energy_rate = a(1:n, 1)
      call check(ctx, sph_download_field(ctx, SPH_F_DALPHA, a(:, 1), n64), 'download dalpha'); bodies%alpha_rate = a(1:n, 1)
    end if
    call check(ctx, sph_get_sinks(ctx, int(ns, c_int32_t), s(:, 1), s(:, 2), s(:, 3), s(:, 4), s(:, 5), s(:, 6), s(:, 7), &
                                  s(:, 8), s(:, 9), s(:, 10)), 'sph_get_sinks')
    do i = 1, ns
      sinks(i)%position = s(i, 1:3)
      sinks(i)%velocity = s(i, 4:6)
      sinks(i)%mass = s(i, 7)
      sinks(i)%acceleration = s(i, 8:10)
    end do
  end subroutine pull_state

  ! ------------------------------------------------------------------------------------------
  ! The time loop.  Per step: density, forces, kick, drift, density, forces, kick, t += dt,
  ! next dt -- one sph_step call.  Saves every end_time/1000 of simulated time.
  ! ------------------------------------------------------------------------------------------
  subroutine simulate(bodies, sinks, end_time_in, max_steps, quiet, device, dt_log, sph_only, saves)
    type(particle), allocatable, intent(inout) :: bodies(:)
    type(sink), intent(inout) :: sinks(:)
    real(dp), intent(in), optional :: end_time_in
    integer, intent(in), optional :: max_steps, device
    logical, intent(in), optional :: quiet
    real(dp), allocatable, intent(out), optional :: dt_log(:)
    logical, intent(in), optional :: sph_only     ! .true.: leave out self-gravity, accretion and the cull
    logical, intent(in), optional :: saves        ! periodic saveN.txt files on / off (default: on unless max_steps is given)

    type(c_ptr) :: ctx
    type(sph_params) :: prm
    real(c_double) :: t, dt
    real(dp) :: end_time, next_save
    real(dp), allocatable :: dts(:)
    integer :: step, save_no, step_limit, dev, i
    logical :: talk, do_saves

    end_time = 1000.0_dp
    if (present(end_time_in)) end_time = end_time_in
    step_limit = huge(1)
    if (present(max_steps)) step_limit = max_steps
    talk = .true.
    if (present(quiet)) talk = .not. quiet
    dev = 0
    if (present(device)) dev = device
    do_saves = .not. present(max_steps)          ! a step-limited (test) run writes no saveN.txt unless asked to
    if (present(saves)) do_saves = saves

    call check(c_null_ptr, sph_params_default(prm), 'sph_params_default')
    prm%h = smoothing
    ! find_forces as it is (with the gas self-gravity term) + end-of-step accretion and cull, [F]:825,919-920
    prm%flags = ior(SPH_FLAG_SELF_GRAVITY, SPH_FLAG_ACCRETE_CULL)
    if (present(sph_only)) then
      if (sph_only) prm%flags = 0
    end if
    prm%bounding_size = bounding_size
    call check(c_null_ptr, sph_ctx_create(prm, int(dev, c_int), ctx), 'sph_ctx_create')
    call push_state(ctx, bodies, sinks)

    t = 0.0_c_double
    dt = 1.0e-2_c_double
    next_save = 0.0_dp      ! the reference writes save0 on its first iteration
    save_no = 0
    step = 0
    allocate(dts(0:min(step_limit, 100000)))
    dts(0) = dt

    do while (t < end_time .and. step < step_limit)
      ! save check: the reference compares t > t_list(t_test) with t_list(i) = i*end_time/1000 and t_test starting at 0
      ! (out of bounds: observed to write save0 on the first iteration); one save per iteration at most
      if (do_saves) then
        if (save_no == 0 .or. t > next_save) then
          call pull_state(ctx, bodies, sinks, .false.)
          call make_save(bodies, sinks, save_no)
          save_no = save_no + 1
          next_save = (save_no * end_time) / 1000
        end if
      end if
      if (talk) print *, 'SPH Particles:', size(bodies), 'dt :', dt, 'time : ', t

      call check(ctx, sph_step(ctx, dt, t), 'sph_step')
      step = step + 1
      if (step <= ubound(dts, 1)) dts(step) = dt

      ! accretion / cull happened on the device: follow the particle count
      if (int(sph_count(ctx)) /= size(bodies)) then
        deallocate(bodies)
        allocate(bodies(int(sph_count(ctx))))
      end if
    end do

    call pull_state(ctx, bodies, sinks, step > 0)
    do i = 1, size(bodies)
      bodies(i)%number = i
    end do
    if (present(dt_log)) then
      allocate(dt_log(0:min(step, ubound(dts, 1))))
      dt_log = dts(0:ubound(dt_log, 1))
    end if
    call check(ctx, sph_ctx_destroy(ctx), 'sph_ctx_destroy')
  end subroutine simulate

end module sph_hip_host
